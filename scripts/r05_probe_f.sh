#!/bin/bash
# round 5, probe F (one box): which of the two capped kernels moves the lower half band's step (probe E: +5.5 us there, -1..-4 us everywhere else)
#   head = no cap (96), capT = k1_tile_cull only, capP = k01_prepare only, default = both at 80
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05q}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
AB=$PWD/sailor_amd/csrc/ab
for rep in 1 2 3; do
for v in head capT capP default; do
    L=$AB/libsailor_hip_$v.so; [ $v = default ] && L=$PWD/sailor_amd/csrc/libsailor_hip.so
    SAILOR_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 48 > $OUT/c3_whole_${v}_$rep.json 2> /dev/null
    for b in 0/2 1/2 0/4; do
        SAILOR_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 48 --simulate-band $b > $OUT/c3_band$(echo $b | tr / o)_${v}_$rep.json 2> /dev/null
    done
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
