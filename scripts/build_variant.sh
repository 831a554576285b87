#!/bin/bash
# Builds libsailor_hip.so from the CURRENT sources with extra compiler flags into sailor_amd/csrc/ab/libsailor_hip_<name>.so (git-ignored, travels with gpurun), for
# same-box A / B runs through SAILOR_HIP_LIB:   scripts/build_variant.sh stats -DRASTER_STATS
set -eu
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/sailor_variant_XXXX)
mkdir -p "$tmp/sailor_amd" "$tmp/include"
cp -r "$root/sailor_amd/csrc" "$tmp/sailor_amd/csrc"
cp "$root/include/sailor_hip.h" "$tmp/include/"
rm -f "$tmp"/sailor_amd/csrc/*.o "$tmp"/sailor_amd/csrc/*.so
make -s -j8 -C "$tmp/sailor_amd/csrc" EXTRA="$*" 2>&1 | grep -E "error|Error" || true
mkdir -p "$root/sailor_amd/csrc/ab"
cp "$tmp/sailor_amd/csrc/libsailor_hip.so" "$root/sailor_amd/csrc/ab/libsailor_hip_$name.so"
rm -rf "$tmp"
echo "built sailor_amd/csrc/ab/libsailor_hip_$name.so ($*)"
