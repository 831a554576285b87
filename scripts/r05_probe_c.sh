#!/bin/bash
# round 5, probe C (one box, one call): the band form's split threshold by band size against the per-tile form; what in k1_pack speeds the following shade up
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05e}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
band() { # name config band extra-env...
    local name=$1 cfg=$2 b=$3; shift 3
    env "$@" python bench.py --config $cfg --simulate-band $b --no-cpu-baseline --steps 48 --static-lights > $OUT/$name.json 2> $OUT/$name.err
}
for b in 0/2 1/2 1/4 3/8; do
    t=$(echo $b | tr / o)
    band c3_${t}_tileform C3 $b SAILOR_BAND_FORM_TILES=0
    for m in 40 64 96 128; do band c3_${t}_band_split$m C3 $b SAILOR_BAND_FORM_TILES=100000 SAILOR_SPLIT_MIN=$m; done
done
band c5_3o8_tileform C5 3/8 SAILOR_BAND_FORM_TILES=0
for m in 40 96 128; do band c5_3o8_band_split$m C5 3/8 SAILOR_BAND_FORM_TILES=100000 SAILOR_SPLIT_MIN=$m; done
band c5_3o8_band_split96_lds0 C5 3/8 SAILOR_BAND_FORM_TILES=100000 SAILOR_SPLIT_MIN=96 SAILOR_BAND_SHADE_LDS=0
python3 scripts/r05_marker_probe.py C3 > $OUT/marker_probe.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/c*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f.split("/")[-1], "unreadable", e); continue
    r = d["roofline"]
    print("%-34s step %6.1f serial %6.1f  %-18s %6.1f  %s" % (f.split("/")[-1], d["ms_per_step"] * 1e3, d["serial_step_ms"]["median"] * 1e3, r["kernel"], r["avg_launch_ms"] * 1e3,
          {k: round(v * 1e3, 1) for k, v in r["cull"]["kernels_ms"].items()}))
PY
tail -6 $OUT/marker_probe.txt
