cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_red gpurun_out/prof_loop
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_red -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/prof_red.log 2>&1
python3 -c "
import csv,glob,os
f=max(glob.glob('gpurun_out/prof_red/*/*_kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:4]: print('bench', r['Name'][:60], r['Calls'], r['AverageNs'])
"
LOOP_ARGS="102400 4" bash scripts/exp/prof_solver_loop.sh 2>&1 | sed -n 2,5p
python3 scripts/solver_loop_rate.py 102400 10 2>&1 | tail -1
