#!/bin/bash
# The shadow passes (1 M entities, four 4096^2 cascades) under settings of the round-6 rasteriser switches, timings on the product library, the rasteriser's own
# counters on the -DRASTER_STATS build: scripts/r06_raster_ab.sh <tag> "<ENV=.. ENV=..>" "<...>" ...  -> gpurun_out/<tag>/raster_<n>[_stats].json + a summary
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
n=0
for setting in "$@"; do
  env $setting python scripts/r06_raster_probe.py 5 > $out/raster_$n.json 2> $out/raster_$n.err
  env $setting SAILOR_HIP_LIB=$PWD/sailor_amd/csrc/ab/libsailor_hip_rstats.so python scripts/r06_raster_probe.py 2 > $out/raster_${n}_stats.json 2> /dev/null
  echo "$setting" > $out/raster_$n.setting
  n=$((n+1))
done
python - $out $n <<'PY'
import json, sys
out, n = sys.argv[1], int(sys.argv[2])
for i in range(n):
    setting = open(f"{out}/raster_{i}.setting").read().strip()
    try:
        d = json.load(open(f"{out}/raster_{i}.json")); s = json.load(open(f"{out}/raster_{i}_stats.json"))
    except Exception as e:
        print(setting, "unreadable", e); continue
    print(f"{setting}: all passes {d['all_passes_ms']:.2f} ms; per cascade " + " / ".join(f"{k['raster_ms']:.2f}" for k in d["cascades"]))
    for k, kk in enumerate(s["cascades"]):
        st = kk.get("stats", {})
        print(f"    cascade {k}: instances {kk['instances']}, lanes of hidden instances {st.get('extra')}, superblocks {st.get('superblocks_alive')}/{st.get('superblocks_seen')}, "
              f"blocks {st.get('blocks_alive')}/{st.get('blocks_seen')}, texels inside {st.get('texels_inside')} written {st.get('texels_written')}; "
              f"waves {st.get('waves')}: mean {st.get('wave_ticks_sum', 0) / max(st.get('waves', 1), 1) / 100:.1f} us, longest {st.get('wave_ticks_max', 0) / 100:.0f} us, "
              f"{st.get('waves_over_100us')} above 100 us, {st.get('waves_over_1ms')} above 1 ms")
PY
