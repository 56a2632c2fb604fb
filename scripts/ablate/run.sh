#!/bin/bash
# timing-only ablation builds of the Q5 kernels (outputs are wrong by construction): which part costs what.
# Build them with `make -C quanonet_amd/csrc ablate`.
cd "$GRAFT_REPO_ROOT"
for v in BASE RING SUMS INNER; do
  for k in packed pair; do
    echo "== $v $k"
    QHEA_BACKWARD_KERNEL=$k QHEA_LIB=$PWD/scripts/ablate/libab_$v.so python scripts/perf_gpu.py cfg2nocheck 2>&1 | grep "n=5 B=1024"
  done
done
