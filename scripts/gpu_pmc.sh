#!/bin/bash
# SQ counters of the C3 frame's kernels (run through gpurun from the repo root): bash scripts/gpu_pmc.sh <tag> <kernel substring> [config]
TAG=${1:-pmc}; PAT=${2:-shade}; CFG=${3:-C3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 -i $GRAFT_REPO_ROOT/scripts/pmc_shade.txt --kernel-trace --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame.py $CFG 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT/sq $PAT > $OUT/pmc_$PAT.txt
rm -rf $OUT/sq
cat $OUT/pmc_$PAT.txt
