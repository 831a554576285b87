#!/bin/bash
# A / B builds of libsailor_hip.so for same-box timing: scripts/ab_build.sh NAME "EXTRA flags" [NAME2 "flags2" ...]
# -> sailor_amd/csrc/ab/libsailor_hip_NAME.so (git-ignored, travels with gpurun); the default build is restored at the end.
# Use: SAILOR_HIP_LIB=sailor_amd/csrc/ab/libsailor_hip_NAME.so python bench.py ...
set -e
cd "$(dirname "$0")/../sailor_amd/csrc"
mkdir -p ab
while [ $# -ge 2 ]; do
    name=$1; flags=$2; shift 2
    touch shade.hip light_cull.hip
    make -j8 EXTRA="$flags" > /dev/null
    cp libsailor_hip.so ab/libsailor_hip_$name.so
    echo "built ab/libsailor_hip_$name.so with EXTRA=$flags"
done
touch shade.hip light_cull.hip
make -j8 > /dev/null
echo "default build restored"
