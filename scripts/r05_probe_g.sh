#!/bin/bash
# round 5, probe G (one box): K3's sixteen PCF taps out of one 6 x 6 window (default build) against the tap-by-tap reads (ab/libsailor_hip_nowin.so): parity, then C4
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05r}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_shade_gpu.py tests/test_split_paths_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
AB=$PWD/sailor_amd/csrc/ab
for rep in 1 2 3; do
for v in nowin default; do
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
    acc[(name[0], name[1])].append((d["ms_per_step"] * 1e3, d["serial_step_ms"]["median"] * 1e3, d["shade_ms"]["median"] * 1e3 if isinstance(d.get("shade_ms"), dict) else -1))
for k in sorted(acc):
    print("%-14s %-8s" % k, " ".join("%6.1f/%6.1f/%6.1f" % v for v in acc[k]))
PY
