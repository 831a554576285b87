#!/bin/bash
# round 5, probe R (one box): C4 split eight ways with the shadowed band kernels' reserve gone (default build); the plain band kernels' reserve on C3 bands at 9 000 (default) / 4 096 / 0 bytes
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05res2}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_shade_gpu.py tests/test_split_paths_gpu.py -m gpu -x -q -k "band or split or eight" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
python bench.py --simulate-split 8 --steps 24 --config C4 > $OUT/split8_C4.json 2> /dev/null
python bench.py --simulate-split 4 --steps 24 --config C4 > $OUT/split4_C4.json 2> /dev/null
for rep in 1 2; do
for v in default 4096 0; do
    E=""; [ $v != default ] && E="SAILOR_BAND_SHADE_LDS=$v"
    for b in 3/8 0/8 6/8; do
        env $E python bench.py --no-cpu-baseline --steps 48 --simulate-band $b > $OUT/c3_band$(echo $b | tr / o)_lds${v}_$rep.json 2> /dev/null
    done
done
done
python - <<PY
import json, glob, collections
for f in sorted(glob.glob("$OUT/split*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], "whole %.1f" % (d["whole_frame_ms"] * 1e3), [(k, round(d[k]["predicted_speedup"], 2), [round(x * 1e3, 1) for x in d[k]["band_ms"]]) for k in ("equal", "balanced", "rebalanced") if k in d])
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/c3*.json")):
    d = json.load(open(f))
    name = f.split("/")[-1].rsplit("_", 2)
    acc[(name[0], name[1])].append((d["ms_per_step"] * 1e3, d["serial_step_ms"]["median"] * 1e3, d["roofline"].get("avg_launch_ms", 0) * 1e3))
for k in sorted(acc):
    print("%-14s %-12s" % k, " ".join("%6.1f/%6.1f/%6.1f" % v for v in acc[k]))
PY
