#!/bin/bash
# Same-box A / B of bench.py over the builds of scripts/build_variant.sh (run through gpurun): scripts/ab_bench.sh [bench args] -- NAME...
# Prints cull / shade / step per build, each build twice in alternation (drift shows up as disagreement between the two rounds).
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
mkdir -p gpurun_out/ab
for round in 1 2; do
    for name in default "$@"; do
        lib=sailor_amd/csrc/libsailor_hip.so
        [ "$name" != default ] && lib=sailor_amd/csrc/ab/libsailor_hip_$name.so
        SAILOR_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline "${args[@]}" > gpurun_out/ab/$name.$round.json 2> gpurun_out/ab/$name.$round.err
        python - "$name" "$round" <<PY
import json, sys
try:
    d = json.load(open("gpurun_out/ab/%s.%s.json" % (sys.argv[1], sys.argv[2]))); r = d["roofline"]
    print("%-12s round %s: step %.4f serial %.4f cull %.2f us shade %.2f us frac %.4f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["serial_step_ms"]["median"], d["cull_ms"] * 1e3, d["shade_ms"] * 1e3, r["frac"]))
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
    done
done
