#!/bin/bash
# round 5, probe K (one box): one pass per cascade present in the wave (default build) against the two copies of the look-up code (ab/libsailor_hip_noext.so)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05v}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_shade_gpu.py tests/test_split_paths_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
AB=$PWD/sailor_amd/csrc/ab
for rep in 1 2 3; do
for v in noext default; do
    L=$AB/libsailor_hip_$v.so; [ $v = default ] && L=$PWD/sailor_amd/csrc/libsailor_hip.so
    SAILOR_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 24 --config C4 > $OUT/c4_whole_${v}_$rep.json 2> /dev/null
    SAILOR_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 24 --config C4 --simulate-band 3/8 > $OUT/c4_band3o8_${v}_$rep.json 2> /dev/null
done
done
python - <<PY
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/c*.json")):
    d = json.load(open(f))
    name = f.split("/")[-1].rsplit("_", 2)
    acc[(name[0], name[1])].append((d["ms_per_step"] * 1e3, d["roofline"].get("avg_launch_ms", 0) * 1e3, d["roofline"].get("back_to_back_launch_ms", 0) * 1e3))
for k in sorted(acc):
    print("%-14s %-8s" % k, " ".join("%6.1f/%6.1f/%6.1f" % v for v in acc[k]))
PY
