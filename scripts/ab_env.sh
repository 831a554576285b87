#!/bin/bash
# Same-box A / B of bench.py over an ENVIRONMENT switch of the default build (run through gpurun): scripts/ab_env.sh VAR A B [bench args]
# Prints step / serial / cull / shade per setting, each twice in alternation.
var=$1; a=$2; b=$3; shift 3
mkdir -p gpurun_out/ab
for round in 1 2; do
    for val in $a $b; do
        env $var=$val python bench.py --no-cpu-baseline "$@" > gpurun_out/ab/env_$val.$round.json 2> gpurun_out/ab/env_$val.$round.err
        python - "$var=$val" "$round" "gpurun_out/ab/env_$val.$round.json" <<PY
import json, sys
try:
    d = json.load(open(sys.argv[3])); r = d["roofline"]
    print("%-24s round %s: step %.4f serial %.4f cull %.2f us shade %.2f us (direct %.2f) frac %.4f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["serial_step_ms"]["median"], d["cull_ms"] * 1e3, d["shade_ms"] * 1e3, r["avg_launch_ms"] * 1e3, r["frac"]))
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
    done
done
