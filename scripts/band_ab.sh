#!/bin/bash
# same-box A / B of builds on bands of an 8-way split: scripts/band_ab.sh NAME...   (libs under sailor_amd/csrc/ab/)
for v in default "$@"; do lib=sailor_amd/csrc/libsailor_hip.so; [ $v != default ] && lib=sailor_amd/csrc/ab/libsailor_hip_$v.so
for b in 0 2 4 6; do SAILOR_HIP_LIB=$PWD/$lib timeout 200 python bench.py --simulate-band $b/8 --no-cpu-baseline --steps 30 --single-mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$v band $b', 'step', round(d['ms_per_step']*1e3,1), 'serial', round(d['serial_step_ms']['median']*1e3,1), 'shade', round(d['shade_ms']*1e3,1), 'b2b', round(r['back_to_back_launch_ms']*1e3,1), {k: round(v*1e3,1) for k,v in r['cull']['kernels_ms'].items()})"; done; done
