#!/bin/bash
# one gpurun call: GPU tests, profiles of the bench command, data-parallel rehearsals on the one GPU
set -u
TAG=${1:-r02a}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.log
bash scripts/collect_profiles.sh $TAG || echo "collect failed"
python3 scripts/summarize_profiles.py $TAG > gpurun_out/summary_$TAG.txt 2>&1; tail -12 gpurun_out/summary_$TAG.txt
# 2 / 4 ranks on the one GPU: the data-parallel step sequence end to end, through the peer-mapped exchange (one kernel: sum over
# ranks + Adam) and, for comparison, through the collective (gloo moves the 19 KB through the host) + Adam launch
for x in peer rccl; do
  QHEA_DP_EXCHANGE=$x timeout -k 10 300 python3 bench.py --gpus 2 --same-device --backend gloo --steps 100 --no-cpu-baseline > gpurun_out/dp2_same_device_${x}_$TAG.json 2> gpurun_out/dp2_same_device_${x}_$TAG.err; echo "dp2 $x rc=$?"; cut -c1-250 gpurun_out/dp2_same_device_${x}_$TAG.json
done
timeout -k 10 300 python3 bench.py --gpus 4 --same-device --backend gloo --steps 100 --no-cpu-baseline > gpurun_out/dp4_same_device_peer_$TAG.json 2> gpurun_out/dp4_same_device_peer_$TAG.err; echo "dp4 peer rc=$?"; cut -c1-250 gpurun_out/dp4_same_device_peer_$TAG.json
timeout -k 10 300 python3 scripts/dp_exchange_cost.py > gpurun_out/dp_exchange_cost_$TAG.json 2> gpurun_out/dp_exchange_cost_$TAG.err; echo "dp exchange cost rc=$?"; cat gpurun_out/dp_exchange_cost_$TAG.json
timeout -k 10 300 python3 scripts/perf_gpu.py cfg1 cfg2 cfg4 cfg5 > gpurun_out/other_configs_$TAG.txt 2>&1; echo "other configs rc=$?"; cat gpurun_out/other_configs_$TAG.txt
timeout -k 10 120 python3 -m torch.distributed.run --standalone --nproc-per-node 1 scripts/allreduce_cost.py > gpurun_out/allreduce_cost_$TAG.json 2> gpurun_out/allreduce_cost_$TAG.err; echo "allreduce rc=$?"; cat gpurun_out/allreduce_cost_$TAG.json
# batch sweep of every n <= 5 kernel variant at cfg 2 circuit, and bench line + HBM counters with ONE pipeline per workgroup (AUTO takes two at this batch)
timeout -k 10 300 python3 scripts/ablate/bsweep_all.py > gpurun_out/bsweep_$TAG.txt 2>&1; echo "bsweep rc=$?"; cat gpurun_out/bsweep_$TAG.txt
timeout -k 10 120 python3 bench.py --no-cpu-baseline --backward-variant ztri > gpurun_out/bench_ztri1_$TAG.json 2>/dev/null; echo "ztri (one pipeline per workgroup) rc=$?"; cut -c1-200 gpurun_out/bench_ztri1_$TAG.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/prof_${TAG}_ztri1/pmc_$grp -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --backward-variant ztri > gpurun_out/prof_${TAG}_ztri1_$grp.log 2>&1 || echo "pass $grp failed"
done
