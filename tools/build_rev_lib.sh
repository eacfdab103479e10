#!/bin/bash
# Build the kernel library of another revision for a same-box A/B:  tools/build_rev_lib.sh <git-rev> <name>
# -> gpurun_dbg/libimmunostruct_hip_<name>.so (git-ignored, travels with gpurun; select it with IMMUNOSTRUCT_LIB=...)
set -e
rev=${1:-HEAD}; name=${2:-ref}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/revlib.XXXXXX)
git -C "$root" archive "$rev" immunostruct_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/immunostruct_amd/csrc" -j8 > "$tmp/build.log" 2>&1 || { tail -20 "$tmp/build.log"; exit 1; }
mkdir -p "$root/gpurun_dbg"
cp "$tmp/immunostruct_amd/csrc/libimmunostruct_hip.so" "$root/gpurun_dbg/libimmunostruct_hip_$name.so"
rm -rf "$tmp"
echo "gpurun_dbg/libimmunostruct_hip_$name.so"
