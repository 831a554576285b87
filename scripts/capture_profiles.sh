#!/bin/bash
# Captures the judged artefacts of a round on the GPU box (run through gpurun from the repo root):
#   kernel stats of the bench command, the two PMC traffic passes, the SQ counter passes of the shade / cull kernels (C3 and the C4 shade),
#   the bench line itself, the C5 kernel stats.   usage: bash scripts/capture_profiles.sh <tag>      -> gpurun_out/<tag>/...
TAG=${1:-cap}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-graph --frames-in-flight 1 > $OUT/bench_eager.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --frames-in-flight 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --frames-in-flight 1 > /dev/null 2>&1
rocprofv3 -i $GRAFT_REPO_ROOT/scripts/pmc_shade.txt --kernel-trace --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame.py C3 2 > /dev/null 2>&1
rocprofv3 -i $GRAFT_REPO_ROOT/scripts/pmc_shade.txt --kernel-trace --output-format csv -d $OUT/sq4 -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame.py C4 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats4 -- python3 $GRAFT_REPO_ROOT/bench.py --config C4 --steps 30 --warmup 5 --no-cpu-baseline --no-graph --frames-in-flight 1 > $OUT/bench_eager_C4.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -- python3 $GRAFT_REPO_ROOT/bench.py --config C5 --steps 20 --warmup 3 --no-cpu-baseline --no-graph --frames-in-flight 1 > $OUT/bench_eager_C5.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python3 scripts/make_traffic_json.py $OUT/fetch $OUT/write $OUT/traffic.json C3 > /dev/null
python3 scripts/pmc_summary.py $OUT/sq k2_shade_p > $OUT/pmc_shade.txt
python3 scripts/pmc_summary.py $OUT/sq tile_cull > $OUT/pmc_tile_cull.txt
python3 scripts/pmc_summary.py $OUT/sq4 k2_shade_csm_p > $OUT/pmc_shade_csm_C4.txt
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/stats4 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_C4.csv
cp $(find $OUT/stats5 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_C5.csv
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --config C4 --no-cpu-baseline > $OUT/bench_C4.json 2> $OUT/bench_C4.err
python3 bench.py --simulate-split 8 --steps 30 > $OUT/simulate_split8.json 2> $OUT/simulate_split8.err
rm -rf $OUT/stats $OUT/stats4 $OUT/stats5 $OUT/fetch $OUT/write $OUT/sq $OUT/sq4
ls -la $OUT
