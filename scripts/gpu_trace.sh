#!/bin/bash
# kernel-trace stats of a few eager frames (scripts/prof_frame.py) -> gpurun_out/<tag>_kernel_stats.csv; usage: scripts/gpu_trace.sh <tag> [config] [passes]   (env passes through)
TAG=${1:-trace}; CFG=${2:-C3}; PASSES=${3:-30}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_$TAG -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame.py $CFG $PASSES > /dev/null 2>&1
cp $(find /tmp/trace_$TAG -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
cut -d, -f1-4 $OUT/${TAG}_kernel_stats.csv | sed 's/(.*)"/"/' | head -12
