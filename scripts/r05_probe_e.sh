#!/bin/bash
# round 5, probe E (one box): the scalar-register cap of k1_tile_cull / k01_prepare (default build) against the build without it (ab/libsailor_hip_head.so), band by band
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05p}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
AB=$PWD/sailor_amd/csrc/ab
for rep in 1 2 3; do
for v in head default; do
    L=$AB/libsailor_hip_$v.so; [ $v = default ] && L=$PWD/sailor_amd/csrc/libsailor_hip.so
    SAILOR_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 48 > $OUT/c3_whole_${v}_$rep.json 2> /dev/null
    for b in 0/2 1/2 1/4 2/4 3/8 0/8; do
        SAILOR_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 48 --simulate-band $b > $OUT/c3_band$(echo $b | tr / o)_${v}_$rep.json 2> /dev/null
    done
    SAILOR_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 24 --config C4 > $OUT/c4_whole_${v}_$rep.json 2> /dev/null
done
done
python - <<PY
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/c*.json")):
    d = json.load(open(f))
    name = f.split("/")[-1].rsplit("_", 2)
    acc[(name[0], name[1])].append(d["ms_per_step"] * 1e3)
for k in sorted(acc):
    print("%-14s %-8s" % k, " ".join("%6.1f" % v for v in acc[k]), "  median %.1f" % sorted(acc[k])[len(acc[k]) // 2])
PY
