#!/bin/bash
# Issue-utilisation counters of the Q5 kernels (counters only, separate passes as the guide prescribes).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_issue
rm -rf $OUT && mkdir -p $OUT
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -- python3 scripts/pmc_gpu.py 1024 5 > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_issue/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'qhea' not in k: continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open('gpurun_out/pmc_issue/summary.json', 'w'), indent=1)
for k, d in out.items():
    print(k)
    for c, v in sorted(d.items()): print(f'   {c:24s} {v:14.0f}')
PY
