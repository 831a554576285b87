#!/bin/bash
# One parametrised same-box A / B: `scripts/r06_ab.sh <tag> <env-var> <value A> <value B> [bench args...]` runs bench.py alternately under VAR=A and VAR=B (two rounds
# each, so that a drift of the box shows), and prints the headline figures of every run.  Output: gpurun_out/<tag>/{A,B}_<round>.json + summary.txt.
set -u
tag=$1; var=$2; a=$3; b=$4; shift 4
out=gpurun_out/$tag
mkdir -p "$out"
for round in 1 2; do
  for side in A B; do
    val=$a; [ $side = B ] && val=$b
    env "$var=$val" python bench.py --no-cpu-baseline "$@" > "$out/${side}_$round.json" 2> "$out/${side}_$round.err"
  done
done
python - "$out" "$var" "$a" "$b" <<'PY' | tee "$out/summary.txt"
import json, sys, glob, os
out, var, a, b = sys.argv[1:5]
for side, val in (("A", a), ("B", b)):
    for f in sorted(glob.glob(f"{out}/{side}_*.json")):
        try:
            d = json.load(open(f))
        except Exception as e:
            print(f"{var}={val} {os.path.basename(f)}: unreadable ({e})"); continue
        r = d["roofline"]
        print(f"{var}={val} {os.path.basename(f)}: step {d['ms_per_step']*1e3:.1f} us  serial {d['serial_step_ms']['median']*1e3:.1f}  {r['kernel']} {r['avg_launch_ms']*1e3:.1f} us  "
              f"back-to-back {r['back_to_back_launch_ms']*1e3:.1f}  cull {sum(r['cull']['kernels_ms'].values())*1e3:.1f}  copy {d['box']['copy_gbs']:.0f} GB/s")
PY
