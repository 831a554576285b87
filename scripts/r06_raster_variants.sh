#!/bin/bash
# The shadow passes on variant builds of the library (scripts/build_variant.sh <name> <flags>): scripts/r06_raster_variants.sh <tag> product <name> <name> ...
# (timing probes such as nofill / nolarge draw WRONG pictures with the right clocks; wtime adds per-wave durations).  gpurun_out/<tag>/
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
for v in "$@"; do
  lib=$PWD/sailor_amd/csrc/libsailor_hip.so; [ $v != product ] && lib=$PWD/sailor_amd/csrc/ab/libsailor_hip_$v.so
  SAILOR_HIP_LIB=$lib python scripts/r06_raster_probe.py 5 > $out/$v.json 2> $out/$v.err
done
python - $out "$@" <<'PY'
import json, sys
out = sys.argv[1]
for v in sys.argv[2:]:
    try:
        d = json.load(open(f"{out}/{v}.json"))
    except Exception as e:
        print(v, "unreadable", e); continue
    print(f"{v}: all passes {d['all_passes_ms']:.2f} ms; per cascade " + " / ".join(f"{k['raster_ms']:.2f}" for k in d["cascades"]))
    for k, kk in enumerate(d["cascades"]):
        st = kk.get("stats")
        if st and st.get("waves"):
            print(f"    cascade {k}: waves {st['waves']}: mean {st['wave_ticks_sum'] / st['waves'] / 100:.1f} us, longest {st['wave_ticks_max'] / 100:.0f} us, "
                  f"{st['waves_over_100us']} above 100 us, {st['waves_over_1ms']} above 1 ms")
PY
