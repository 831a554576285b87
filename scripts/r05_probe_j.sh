#!/bin/bash
# round 5, probe J (one box): where K3's time goes on C4 -- the kernel without its PCF look-ups, without its EVSM look-ups, with every wave treated as inside one
# cascade (results wrong in all three: timing only), beside the window's taps as they are (noext)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05u}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
AB=$PWD/sailor_amd/csrc/ab
for rep in 1 2; do
for v in noext nopcf noevsm alluni; do
    SAILOR_HIP_LIB=$AB/libsailor_hip_$v.so python bench.py --no-cpu-baseline --steps 24 --config C4 > $OUT/c4_whole_${v}_$rep.json 2> /dev/null
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
