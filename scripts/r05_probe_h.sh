#!/bin/bash
# round 5, probe H (one box): what bounds K3's window path -- the six row reads of a pixel from ONE row (one line a pixel instead of six), and no reads at all
# (results are wrong in both: timing only)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05s}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
AB=$PWD/sailor_amd/csrc/ab
for rep in 1 2; do
for v in default samerow noload nowin; do
    L=$AB/libsailor_hip_$v.so; [ $v = default ] && L=$PWD/sailor_amd/csrc/libsailor_hip.so
    SAILOR_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 24 --config C4 > $OUT/c4_whole_${v}_$rep.json 2> /dev/null
done
done
python - <<PY
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/c*.json")):
    d = json.load(open(f))
    name = f.split("/")[-1].rsplit("_", 2)
    acc[(name[0], name[1])].append((d["ms_per_step"] * 1e3, d["roofline"]["avg_launch_ms"] * 1e3, d["roofline"]["back_to_back_launch_ms"] * 1e3))
for k in sorted(acc):
    print("%-14s %-8s" % k, " ".join("%6.1f/%6.1f/%6.1f" % v for v in acc[k]))
PY
