#!/bin/bash
# A variant build of the working tree's kernel library for a same-box A/B:  tools/build_variant_lib.sh <name> "<extra hipcc flags>"
# -> gpurun_dbg/libimmunostruct_hip_<name>.so (git-ignored, travels with gpurun; select it with IMMUNOSTRUCT_LIB=...)
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/varlib.XXXXXX)
cp "$root"/immunostruct_amd/csrc/*.hip "$root"/immunostruct_amd/csrc/*.h "$root"/immunostruct_amd/csrc/Makefile "$tmp"/
make -C "$tmp" -j8 EXTRA="$flags" > "$tmp/build.log" 2>&1 || { tail -20 "$tmp/build.log"; exit 1; }
mkdir -p "$root/gpurun_dbg"
cp "$tmp/libimmunostruct_hip.so" "$root/gpurun_dbg/libimmunostruct_hip_$name.so"
rm -rf "$tmp"
echo "gpurun_dbg/libimmunostruct_hip_$name.so"
