cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_loop
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_loop -- python3 scripts/solver_loop_rate.py ${LOOP_ARGS:-102400 3} > gpurun_out/prof_loop.log 2>&1
tail -1 gpurun_out/prof_loop.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_loop/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['TotalDurationNs'])
PY
python3 - <<'PY'
# host-side cost per step: the same loop body with the GPU work queued but never waited for is GPU-bound, so time the
# pieces on their own
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from quanonet_amd import _lib
x = torch.zeros(102400, 100, dtype=torch.float64, device='cuda'); tails = torch.zeros(100, 2, dtype=torch.float64, device='cuda')
flat = torch.zeros(2403, dtype=torch.float64, device='cuda')
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2000):
    a = x[i:i + 1024]; b = x[i:i + 1024]; c = x[i:i + 1024]
t1 = time.perf_counter()
for i in range(2000):
    tails[i % 100].copy_(flat[2401:])
torch.cuda.synchronize()
t2 = time.perf_counter()
print('3 slices us', (t1 - t0) / 2000 * 1e6, 'tail copy us (incl. gpu)', (t2 - t1) / 2000 * 1e6)
PY
