#!/bin/bash
# round 5, probe A (one box, one call): the band forms by band size, the wide path's cluster threshold, the 8-way split without k1_pack
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05b}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
band() { # name config band extra-env...
    local name=$1 cfg=$2 b=$3; shift 3
    env "$@" python bench.py --config $cfg --simulate-band $b --no-cpu-baseline --steps 48 --static-lights > $OUT/$name.json 2> $OUT/$name.err
}
for b in 0/2 1/4; do
    t=$(echo $b | tr / o)
    band c3_${t}_bandform C3 $b
    band c3_${t}_tileform C3 $b SAILOR_BAND_FORM_TILES=6000
    band c3_${t}_tileform_lds0 C3 $b SAILOR_BAND_FORM_TILES=6000 SAILOR_BAND_SHADE_LDS=0
    band c3_${t}_tileform_lds9000 C3 $b SAILOR_BAND_FORM_TILES=6000 SAILOR_BAND_SHADE_LDS=9000
done
band c5_3o8_bandform C5 3/8
band c5_3o8_bandform_heavy0 C5 3/8 SAILOR_HEAVY_MIN_WIDE=0
band c5_3o8_bandform_heavy768 C5 3/8 SAILOR_HEAVY_MIN_WIDE=768
band c5_3o8_bandform_heavy1536 C5 3/8 SAILOR_HEAVY_MIN_WIDE=1536
band c5_3o8_tileform C5 3/8 SAILOR_BAND_FORM_TILES=6000
band c5_3o8_tileform_lds0 C5 3/8 SAILOR_BAND_FORM_TILES=6000 SAILOR_BAND_SHADE_LDS=0
python bench.py --simulate-split 8 --steps 30 > $OUT/simulate_split8.json 2> $OUT/simulate_split8.err
python bench.py --simulate-split 8 --steps 30 --pack deferred > $OUT/simulate_split8_pack_deferred.json 2> $OUT/simulate_split8_pack_deferred.err
python bench.py --no-cpu-baseline --steps 48 > $OUT/bench_nocpu.json 2> $OUT/bench_nocpu.err
python bench.py --no-cpu-baseline --steps 48 --pack deferred > $OUT/bench_nocpu_pack_deferred.json 2> $OUT/bench_nocpu_pack_deferred.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f.split("/")[-1], "unreadable", e); continue
    if "roofline" in d:
        r = d["roofline"]
        print("%-34s step %6.1f serial %6.1f  %-18s %6.1f  %s" % (f.split("/")[-1], d["ms_per_step"] * 1e3, d["serial_step_ms"]["median"] * 1e3, r["kernel"], r["avg_launch_ms"] * 1e3,
              {k: round(v * 1e3, 1) for k, v in r["cull"]["kernels_ms"].items()}))
    else:
        print(f.split("/")[-1], {k: (round(v["predicted_speedup"], 2), [round(x * 1e3, 1) for x in v["band_ms"]]) for k, v in d.items() if isinstance(v, dict) and "band_ms" in v}, round(d["whole_frame_ms"] * 1e3, 1))
PY
