#!/bin/bash
# Runs on the GPU box: the TorchQuantum-shaped baseline on the MI355X (eager PyTorch-ROCm, complex128 and complex64)
# and on the host CPU, and the C oracle with one thread and with all threads.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/baselines.jsonl
: > $OUT
timeout -k 10 300 python3 scripts/tq_shaped_baseline.py --device cuda --dtype c128 --steps 5 --check >> $OUT 2>gpurun_out/baselines.err
timeout -k 10 300 python3 scripts/tq_shaped_baseline.py --device cuda --dtype c64 --steps 5 >> $OUT 2>>gpurun_out/baselines.err
timeout -k 10 300 python3 scripts/tq_shaped_baseline.py --device cpu --dtype c128 --steps 2 >> $OUT 2>>gpurun_out/baselines.err
for t in 1 0; do
  if [ $t = 1 ]; then export OMP_NUM_THREADS=1; else unset OMP_NUM_THREADS; fi
  timeout -k 10 300 python3 - >> $OUT 2>>gpurun_out/baselines.err <<'PY'
import json, time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import c_oracle as C, hea_oracle as O
n, net = 5, (40, 2, 20, 2)
cfgs = O.block_configs_quanonet(n, net); E, blk = O.circuit_sizes(n, cfgs)
rng = np.random.default_rng(0)
th = C.threads(); nb = 64 * max(1, th // 2)
x = rng.uniform(-3, 3, (nb, E)); w = rng.uniform(-3, 3, (blk, 3, n)); g = rng.normal(size=nb)
off, co = O.ham_params(n)
C.hea_backward(n, cfgs, x[:8], w, g[:8], off, co)
t0 = time.perf_counter(); done = 0
while time.perf_counter() - t0 < 8: C.hea_backward(n, cfgs, x, w, g, off, co); done += nb
dt = time.perf_counter() - t0
t1 = time.perf_counter(); fd = 0
while time.perf_counter() - t1 < 4: C.hea_forward(n, cfgs, x, w, off, co); fd += nb
dtf = time.perf_counter() - t1
print(json.dumps({"baseline": "oracle/hea_oracle.c (C + OpenMP over the batch)", "threads": th,
                  "train_samples_per_s": done / dt, "forward_evals_per_s": fd / dtf}))
PY
done
cat $OUT
lscpu | grep "Model name\|^CPU(s)" | head -2
