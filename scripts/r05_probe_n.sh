#!/bin/bash
# round 5, probe N (one box): the cull stream's priority in the frame pipeline (SAILOR_CULL_PRIORITY: torch stream priority, lower = more urgent), whole frame and bands
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05pri}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)" 2>/dev/null
for rep in 1 2; do
for pr in 0 -1 1; do
    SAILOR_CULL_PRIORITY=$pr python bench.py --no-cpu-baseline --steps 48 > $OUT/c3_whole_p${pr}_$rep.json 2> /dev/null
    SAILOR_CULL_PRIORITY=$pr python bench.py --no-cpu-baseline --steps 48 --simulate-band 3/8 > $OUT/c3_band3o8_p${pr}_$rep.json 2> /dev/null
    SAILOR_CULL_PRIORITY=$pr python bench.py --no-cpu-baseline --steps 48 --simulate-band 1/4 > $OUT/c3_band1o4_p${pr}_$rep.json 2> /dev/null
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
    print("%-14s %-6s" % k, " ".join("%6.1f" % v for v in acc[k]))
PY
