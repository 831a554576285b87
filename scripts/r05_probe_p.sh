#!/bin/bash
# round 5, probe P (one box): the band form's launch order refreshed BEHIND the band's shade (default: the next shade goes by the lengths of the frame before)
# against tile order (SAILOR_NO_LAUNCH_ORDER=1)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05ord4}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_shade_gpu.py tests/test_split_paths_gpu.py tests/test_runtime_gpu.py tests/test_bench_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for rep in 1 2; do
for o in on off; do
    E=""; [ $o = off ] && E="SAILOR_NO_LAUNCH_ORDER=1"
    for b in 3/8 0/8 1/4 2/4 0/2 1/2; do
        env $E python bench.py --no-cpu-baseline --steps 48 --simulate-band $b > $OUT/c3_band$(echo $b | tr / o)_${o}_$rep.json 2> /dev/null
    done
    env $E python bench.py --no-cpu-baseline --steps 24 --config C4 --simulate-band 3/8 > $OUT/c4_band3o8_${o}_$rep.json 2> /dev/null
    env $E python bench.py --no-cpu-baseline --steps 24 --config C4 --simulate-band 6/8 > $OUT/c4_band6o8_${o}_$rep.json 2> /dev/null
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
    print("%-14s %-4s" % k, " ".join("%6.1f/%6.1f/%6.1f" % v for v in acc[k]))
PY
