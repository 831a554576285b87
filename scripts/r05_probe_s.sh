#!/bin/bash
# round 5, probe S (one box): the split threshold of the band form on C4's eighth bands (SAILOR_SPLIT_MIN; default 64 up to 12 000 tiles)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05spl}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in default 40 48 80 96; do
    E=""; [ $v != default ] && E="SAILOR_SPLIT_MIN=$v"
    for b in 3/8 6/8 0/8; do
        env $E python bench.py --no-cpu-baseline --steps 24 --config C4 --simulate-band $b > $OUT/c4_band$(echo $b | tr / o)_min${v}_$rep.json 2> /dev/null
    done
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
    print("%-14s %-12s" % k, " ".join("%6.1f/%6.1f/%6.1f" % v for v in acc[k]))
PY
