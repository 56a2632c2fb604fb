#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats of the same command, and the two
# HBM-traffic PMC passes (separate runs, counters only) as MI355X_MICROARCH.md prescribes.
# Usage: scripts/collect_profiles.sh <tag>
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 300 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
echo "bench done"; cut -c1-200 "$OUT/bench.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/trace.log" 2>&1 || exit 1
echo "trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1 || exit 1
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1 || exit 1
echo "write done"
