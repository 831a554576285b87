#!/bin/bash
# Quick GPU iteration (run through gpurun from the repo root): parity tests of the given files, then the per-kernel
# times of the eager C3 frame under rocprofv3.   usage: bash scripts/gpu_quick.sh <tag> [pytest args...]
TAG=${1:-quick}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
if [ $# -gt 0 ]; then timeout 1200 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -15; fi
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-graph --frames-in-flight 1 --single-mode ${BENCH_ARGS} > $OUT/bench_eager.json 2> $OUT/bench_eager.err
cd $GRAFT_REPO_ROOT
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows[:9]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us  x{r['Calls']:>5}  {r['Name'][:70]}")
PY
