#!/bin/bash
# round 5, probe D (one box): candidate records beside the group lists, and the scalar-register cap (eight blocks per CU) of k1_tile_cull / k01_prepare
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05n}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
AB=$PWD/sailor_amd/csrc/ab
for rep in 1 2; do
for v in head default; do
    L=$AB/libsailor_hip_$v.so; [ $v = default ] && L=$PWD/sailor_amd/csrc/libsailor_hip.so
    SAILOR_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 48 > $OUT/c3_${v}_$rep.json 2> /dev/null
    SAILOR_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 48 --simulate-band 3/8 > $OUT/c3_band3o8_${v}_$rep.json 2> /dev/null
    SAILOR_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 48 --simulate-band 0/2 > $OUT/c3_band0o2_${v}_$rep.json 2> /dev/null
    SAILOR_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 24 --config C5 --static-lights --simulate-band 3/8 > $OUT/c5_band3o8_${v}_$rep.json 2> /dev/null
done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/c*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print("%-28s step %6.1f serial %6.1f  %-18s %6.1f  %s" % (f.split("/")[-1], d["ms_per_step"] * 1e3, d["serial_step_ms"]["median"] * 1e3, r["kernel"], r["avg_launch_ms"] * 1e3,
          {k: round(v * 1e3, 1) for k, v in r["cull"]["kernels_ms"].items()}))
PY
