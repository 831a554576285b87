#!/bin/bash
# Captures the judged artefacts of a round on the GPU box (run through gpurun from the repo root):
#   kernel stats of the bench command (C3, C4, C5), the PMC traffic passes (C3 AND C4), the SQ counter passes of the shade / cull kernels (C3 and the C4
#   shade), the bench lines (C3, C4, C5), the split simulations (2 / 4 / 8 bands), the frame pipeline's kernel timelines (whole frame; one band).
#   (libsailor_hip_prof.so: scripts/build_variant.sh prof "-DCULL_PROF -DSHADE_PROF", built HERE in the container before the call: the .so travels)
#   usage: bash scripts/capture_profiles.sh <tag>      -> gpurun_out/<tag>/...   (copy what is to be judged into profiles/<round>/)
TAG=${1:-cap}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
EAGER="--no-cpu-baseline --no-graph --frames-in-flight 1 --single-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B --steps 50 --warmup 5 $EAGER > $OUT/bench_eager.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $B --steps 5 --warmup 2 $EAGER > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $B --steps 5 --warmup 2 $EAGER > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch4 -- python3 $B --config C4 --steps 5 --warmup 2 $EAGER > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write4 -- python3 $B --config C4 --steps 5 --warmup 2 $EAGER > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch5 -- python3 $B --config C5 --steps 3 --warmup 1 $EAGER > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write5 -- python3 $B --config C5 --steps 3 --warmup 1 $EAGER > /dev/null 2>&1
rocprofv3 -i $GRAFT_REPO_ROOT/scripts/pmc_shade.txt --kernel-trace --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame.py C3 2 > /dev/null 2>&1
rocprofv3 -i $GRAFT_REPO_ROOT/scripts/pmc_shade.txt --kernel-trace --output-format csv -d $OUT/sq4 -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame.py C4 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats4 -- python3 $B --config C4 --steps 30 --warmup 5 $EAGER > $OUT/bench_eager_C4.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -- python3 $B --config C5 --steps 20 --warmup 3 $EAGER > $OUT/bench_eager_C5.json 2>/dev/null
# the frame pipeline as the default line launches it (hipGraph, two frames in flight): the kernels' start / end times inside it
rocprofv3 --kernel-trace --output-format csv -d $OUT/pipe -- python3 $B --steps 48 --warmup 6 --no-cpu-baseline --single-mode > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pipeband -- python3 $B --steps 48 --warmup 6 --no-cpu-baseline --single-mode --simulate-band 2/8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/make_traffic_json.py $OUT/fetch $OUT/write $OUT/traffic.json C3 > /dev/null
python3 scripts/make_traffic_json.py $OUT/fetch4 $OUT/write4 $OUT/traffic_C4.json C4 > /dev/null
python3 scripts/make_traffic_json.py $OUT/fetch5 $OUT/write5 $OUT/traffic_C5.json C5 > /dev/null
# (round 6) C5's wide list builder in round 5's form (one row of four groups per block), for the traffic and time beside the 4 x 4-patch form above
cd /tmp
SAILOR_CULL_WIDE16=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch5w -- python3 $B --config C5 --steps 3 --warmup 1 $EAGER > /dev/null 2>&1
SAILOR_CULL_WIDE16=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write5w -- python3 $B --config C5 --steps 3 --warmup 1 $EAGER > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/make_traffic_json.py $OUT/fetch5w $OUT/write5w $OUT/traffic_C5_wide16_off.json C5 > /dev/null
for rep in 1 2; do for v in 1 0; do SAILOR_CULL_WIDE16=$v python3 bench.py --config C5 --no-cpu-baseline --steps 20 > $OUT/bench_C5_wide16_${v}_$rep.json 2> /dev/null; done; done
python3 scripts/r06_raster_probe.py 5 > $OUT/shadow_passes.json 2> /dev/null
python3 scripts/pmc_summary.py $OUT/sq k2_shade_pt > $OUT/pmc_shade.txt
python3 scripts/pmc_summary.py $OUT/sq tile_cull > $OUT/pmc_tile_cull.txt
python3 scripts/pmc_summary.py $OUT/sq4 k2_shade_csm_pt > $OUT/pmc_shade_csm_C4.txt
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/stats4 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_C4.csv
cp $(find $OUT/stats5 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_C5.csv
python3 scripts/analysis/pipeline_timeline.py $(find $OUT/pipe -name "*kernel_trace.csv" | head -1) > $OUT/pipeline_timeline_C3.txt 2>&1
python3 scripts/analysis/pipeline_timeline.py $(find $OUT/pipeband -name "*kernel_trace.csv" | head -1) > $OUT/pipeline_timeline_band2of8.txt 2>&1
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --config C4 --no-cpu-baseline > $OUT/bench_C4.json 2> $OUT/bench_C4.err
python3 bench.py --config C5 --no-cpu-baseline --steps 20 > $OUT/bench_C5.json 2> $OUT/bench_C5.err
# N > 1 as the driver types it, on this ONE-GPU box: the ranks share device 0 over gloo (figures meaningless; the path and the exchanged lists are real)
for G in 2 4; do SAILOR_BENCH_SHARE_GPU=1 python3 bench.py --gpus $G --steps 12 --no-cpu-baseline > $OUT/bench_${G}ranks_sharing_one_gpu.json 2> $OUT/bench_${G}ranks_sharing_one_gpu.err; done
for G in 2 4 8; do python3 bench.py --simulate-split $G --steps 30 > $OUT/simulate_split$G.json 2> $OUT/simulate_split$G.err; done
# the 8-GPU configurations of BASELINE.json (configs[3], configs[4]) band by band: C4 with its shadow maps, C5 static and with every light dirty every frame
python3 bench.py --simulate-split 8 --steps 24 --config C4 > $OUT/simulate_split8_C4.json 2> /dev/null
python3 bench.py --simulate-split 8 --steps 24 --config C5 --static-lights > $OUT/simulate_split8_C5.json 2> /dev/null
python3 bench.py --simulate-split 8 --steps 24 --config C5 > $OUT/simulate_split8_C5_dynamic.json 2> /dev/null
# a rank's line of an 8-way, a 4-way and a 2-way split, kernel by kernel (dispatch-packet readings): C3 bands and a C5 band
for b in 0/2 1/2 1/4 2/4 3/8; do python3 bench.py --simulate-band $b --no-cpu-baseline --steps 48 > $OUT/bench_C3_band$(echo $b | tr / of).json 2> /dev/null; done
python3 bench.py --config C5 --simulate-band 3/8 --no-cpu-baseline --steps 24 --static-lights > $OUT/bench_C5_band3of8_static.json 2> /dev/null
python3 bench.py --config C5 --simulate-band 3/8 --no-cpu-baseline --steps 24 > $OUT/bench_C5_band3of8_dynamic.json 2> /dev/null
# block timelines (prof build): the shade of half / a quarter / an eighth of the 4K frame in the band form, the whole frame and a C5 band on the per-tile grid; per-wave slots
PROF=$PWD/sailor_amd/csrc/ab/libsailor_hip_prof.so
SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof.py 0/2 > $OUT/shade_block_timeline_C3_band0of2.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof.py 1/4 > $OUT/shade_block_timeline_C3_band1of4.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof.py 2/8 > $OUT/shade_block_timeline_C3_band2of8.txt 2>&1
SAILOR_BAND_FORM_TILES=0 SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof_grid.py 0/2 C3 > $OUT/shade_block_timeline_C3_band0of2_per_tile_grid.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof_grid.py 0/1 C3 > $OUT/shade_block_timeline_C3_whole.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof_grid.py 3/8 C5 > $OUT/shade_block_timeline_C5_band3of8.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/shade_wave_prof.py 0/1 C3 > $OUT/shade_waves_C3_whole.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof.py 2/8 C4 > $OUT/shade_block_timeline_C4_band2of8.txt 2>&1
python3 scripts/r05_marker_probe.py C3 > $OUT/shade_behind_pack_marker_probe.txt 2>&1
rm -rf $OUT/stats $OUT/stats4 $OUT/stats5 $OUT/fetch $OUT/write $OUT/fetch4 $OUT/write4 $OUT/fetch5 $OUT/write5 $OUT/fetch5w $OUT/write5w $OUT/sq $OUT/sq4 $OUT/pipe $OUT/pipeband
ls -la $OUT
