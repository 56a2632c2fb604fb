#!/bin/bash
# Runs on the GPU box (via gpurun): the bench line, the rocprofv3 kernel-trace summary of the same command, and the
# PMC passes of the same command -- counters only, one pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit
# one pass; SQ groups of <= 3), as MI355X_MICROARCH.md (HBM / rocprofv3 PMC sections) prescribes.
# Usage: scripts/collect_profiles.sh <tag> [extra bench.py args]
set -u
TAG=${1:-r02}
shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 500 python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
echo "bench done"; cut -c1-300 "$OUT/bench.json"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-secondary "$@" > "$OUT/bench_k20.json" 2>> "$OUT/bench.err" || exit 1
echo "bench k20 done"; cut -c1-200 "$OUT/bench_k20.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --budget-scale 0.2 "$@" > "$OUT/trace.log" 2>&1 || exit 1
echo "trace done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$tag" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --budget-scale 0.05 "$@" > "$OUT/pmc_$tag.log" 2>&1 || echo "pass $tag failed"
  echo "pmc $tag done"
done
