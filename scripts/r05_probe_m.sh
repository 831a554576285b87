#!/bin/bash
# round 5, probe M (one box): C4's eight bands -- the shadowed band kernels at six (default) and seven waves per SIMD, and the whole frame's launch form on the bands
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05y}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_bench_gpu.py -m gpu -x -q > $OUT/pytest_bench.log 2>&1; tail -2 $OUT/pytest_bench.log
AB=$PWD/sailor_amd/csrc/ab
python bench.py --simulate-split 8 --steps 24 --config C4 > $OUT/split8_C4_default.json 2> /dev/null
SAILOR_HIP_LIB=$AB/libsailor_hip_bandw7.so python bench.py --simulate-split 8 --steps 24 --config C4 > $OUT/split8_C4_bandw7.json 2> /dev/null
SAILOR_BAND_FORM_TILES=0 python bench.py --simulate-split 8 --steps 24 --config C4 > $OUT/split8_C4_tileform.json 2> /dev/null
SAILOR_HIP_LIB=$AB/libsailor_hip_bandw7.so python bench.py --simulate-split 4 --steps 24 --config C4 > $OUT/split4_C4_bandw7.json 2> /dev/null
python bench.py --simulate-split 4 --steps 24 --config C4 > $OUT/split4_C4_default.json 2> /dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/split*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], "whole %.1f" % (d["whole_frame_ms"] * 1e3), [(k, round(d[k]["predicted_speedup"], 2), [round(x * 1e3, 1) for x in d[k]["band_ms"]]) for k in ("equal", "balanced", "rebalanced") if k in d])
PY
