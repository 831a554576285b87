#!/bin/bash
# Per-kernel times of ONE band of a G-way split of the C3 frame, eager launches, serial frames (run through gpurun from the repo
# root): bash scripts/gpu_band.sh <tag> <band R/G> [more bands...]
TAG=${1:-band}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
  N=${B//\//of}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$N -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-graph --frames-in-flight 1 --simulate-band $B ${BENCH_ARGS} > $OUT/band_$N.json 2> $OUT/band_$N.err
  cp $(find $OUT/stats_$N -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$N.csv
  rm -rf $OUT/stats_$N
  echo "== band $B"
  python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$OUT/kernel_stats_$N.csv")))
tot = 0.0
for r in rows[:7]:
    if int(r['Calls']) > 10:
        tot += float(r['AverageNs'])/1e3
    print(f"{float(r['AverageNs'])/1e3:9.1f} us  x{r['Calls']:>5}  {r['Name'][:60]}")
print(f"   sum of per-step kernels {tot:.1f} us")
try:
    j = json.loads(open("$OUT/band_$N.json").read().strip().splitlines()[-1]); print("   ms_per_step", j.get("ms_per_step"))
except Exception as e: print("   (no json)", e)
PY
done
