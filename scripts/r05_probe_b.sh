#!/bin/bash
# round 5, probe B (one box, one call): block timelines of half / quarter bands and of a C5 band (prof build), then the split simulations with the new defaults
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05d}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
PROF=$PWD/sailor_amd/csrc/ab/libsailor_hip_prof.so
SAILOR_BAND_FORM_TILES=100000 SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof.py 0/2 > $OUT/shade_blocks_C3_band0of2_bandform.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof_grid.py 0/2 C3 > $OUT/shade_blocks_C3_band0of2_tileform.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof.py 1/4 > $OUT/shade_blocks_C3_band1of4_bandform.txt 2>&1
SAILOR_BAND_FORM_TILES=0 SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof_grid.py 1/4 C3 > $OUT/shade_blocks_C3_band1of4_tileform.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof_grid.py 0/1 C3 > $OUT/shade_blocks_C3_whole.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof_grid.py 3/8 C5 > $OUT/shade_blocks_C5_band3of8_tileform.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/cull_prof.py 0/2 > $OUT/cull_blocks_C3_band0of2.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/cull_prof.py 1/4 > $OUT/cull_blocks_C3_band1of4.txt 2>&1
SAILOR_HIP_LIB=$PROF python3 scripts/cull_prof.py 3/8 C5 > $OUT/cull_blocks_C5_band3of8.txt 2>&1
for G in 2 4 8; do python3 bench.py --simulate-split $G --steps 30 > $OUT/simulate_split$G.json 2> $OUT/simulate_split$G.err; done
python3 bench.py --simulate-split 8 --steps 24 --config C4 > $OUT/simulate_split8_C4.json 2> $OUT/simulate_split8_C4.err
python3 bench.py --simulate-split 8 --steps 24 --config C5 --static-lights > $OUT/simulate_split8_C5.json 2> $OUT/simulate_split8_C5.err
python3 bench.py --simulate-split 8 --steps 24 --config C5 > $OUT/simulate_split8_C5_dynamic.json 2> $OUT/simulate_split8_C5_dynamic.err
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/simulate*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f.split("/")[-1], "unreadable", e); continue
    print(f.split("/")[-1], round(d["whole_frame_ms"] * 1e3, 1), {k: (round(v["predicted_speedup"], 2), v["bounds"], [round(x * 1e3, 1) for x in v["band_ms"]]) for k, v in d.items() if isinstance(v, dict) and "band_ms" in v})
PY
head -12 $OUT/shade_blocks_C3_band0of2_tileform.txt
