#!/bin/bash
# round 5, probe I (one box): K3's PCF -- tap by tap (nowin), the 6 x 6 window's taps (noext), the window's extremes deciding all sixteen compares first with
# three (default) or six (batch6) row reads in flight: parity of the default build, then C4 whole and a band
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05t}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_shade_gpu.py tests/test_split_paths_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
AB=$PWD/sailor_amd/csrc/ab
for rep in 1 2 3; do
for v in nowin noext batch6 default; do
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
