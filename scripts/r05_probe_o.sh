#!/bin/bash
# round 5, probe O (one box): the band form's tile blocks longest list first (k1_band_order, default) against raster order (SAILOR_BAND_ORDER=0)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05ord}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_shade_gpu.py tests/test_split_paths_gpu.py tests/test_runtime_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for rep in 1 2; do
for o in 1 0; do
    for b in 3/8 0/8 1/4 0/2 1/2; do
        SAILOR_BAND_ORDER=$o python bench.py --no-cpu-baseline --steps 48 --simulate-band $b > $OUT/c3_band$(echo $b | tr / o)_order${o}_$rep.json 2> /dev/null
    done
    SAILOR_BAND_ORDER=$o python bench.py --no-cpu-baseline --steps 24 --config C4 --simulate-band 3/8 > $OUT/c4_band3o8_order${o}_$rep.json 2> /dev/null
    SAILOR_BAND_ORDER=$o python bench.py --no-cpu-baseline --steps 24 --config C4 --simulate-band 6/8 > $OUT/c4_band6o8_order${o}_$rep.json 2> /dev/null
done
done
python - <<PY
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/c*.json")):
    d = json.load(open(f))
    name = f.split("/")[-1].rsplit("_", 2)
    acc[(name[0], name[1])].append((d["ms_per_step"] * 1e3, d["serial_step_ms"]["median"] * 1e3, d["roofline"].get("avg_launch_ms", 0) * 1e3))
for k in sorted(acc):
    print("%-14s %-8s" % k, " ".join("%6.1f/%6.1f/%6.1f" % v for v in acc[k]))
PY
PROF=$PWD/sailor_amd/csrc/ab/libsailor_hip_prof.so
SAILOR_HIP_LIB=$PROF python3 scripts/shade_prof.py 2/8 > $OUT/shade_block_timeline_C3_band2of8_ordered.txt 2>&1
head -5 $OUT/shade_block_timeline_C3_band2of8_ordered.txt
