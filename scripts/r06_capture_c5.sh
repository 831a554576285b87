#!/bin/bash
# The C5 pieces of scripts/capture_profiles.sh again (round 6: the capture ran with the 4 x 4-patch list builder on; it is off by default since)
TAG=${1:-r06c5}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
EAGER="--no-cpu-baseline --no-graph --frames-in-flight 1 --single-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -- python3 $B --config C5 --steps 20 --warmup 3 $EAGER > $OUT/bench_eager_C5.json 2>/dev/null
cd $GRAFT_REPO_ROOT
cp $(find $OUT/stats5 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_C5.csv; rm -rf $OUT/stats5
python3 bench.py --config C5 --no-cpu-baseline --steps 20 > $OUT/bench_C5.json 2> $OUT/bench_C5.err
python3 bench.py --simulate-split 8 --steps 24 --config C5 --static-lights > $OUT/simulate_split8_C5.json 2> /dev/null
python3 bench.py --simulate-split 8 --steps 24 --config C5 > $OUT/simulate_split8_C5_dynamic.json 2> /dev/null
python3 bench.py --config C5 --simulate-band 3/8 --no-cpu-baseline --steps 24 --static-lights > $OUT/bench_C5_band3of8_static.json 2> /dev/null
python3 bench.py --config C5 --simulate-band 3/8 --no-cpu-baseline --steps 24 > $OUT/bench_C5_band3of8_dynamic.json 2> /dev/null
SAILOR_HIP_LIB=$PWD/sailor_amd/csrc/ab/libsailor_hip_prof.so python3 scripts/shade_prof_grid.py 3/8 C5 > $OUT/shade_block_timeline_C5_band3of8.txt 2>&1
python3 scripts/r06_raster_probe.py 5 > $OUT/shadow_passes.json 2> /dev/null
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
ls $OUT
