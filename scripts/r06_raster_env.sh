#!/bin/bash
# The shadow passes under a list of environment settings, one probe run each: scripts/r06_raster_env.sh <tag> "<ENV=.. ENV=..>" ...  -> one line per setting
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; n=0
for setting in "$@"; do
  env $setting python scripts/r06_raster_probe.py 5 > $out/run_$n.json 2> $out/run_$n.err
  python -c "
import json,sys
try:
    d=json.load(open('$out/run_$n.json')); print('$setting:', round(d['all_passes_ms'],2), 'ms', [round(k['raster_ms'],2) for k in d['cascades']])
except Exception as e: print('$setting: unreadable', e)"
  n=$((n+1))
done
