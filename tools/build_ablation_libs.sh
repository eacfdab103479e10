#!/bin/bash
# TIMING-ONLY ablation builds of the two layer kernels (results wrong on purpose; HISTORY.md "issue-bound model: ablations"):
#   tools/build_ablation_libs.sh [names...]   -> gpurun_dbg/libimmunostruct_hip_abl_<name>.so  (git-ignored, travel with gpurun)
#   silu   : SiLU / SiLU' -> one multiply / a constant (no v_exp / v_rcp, no sigma arithmetic)
#   seg    : destination segment sums skipped (forward: the scan + flushes; backward: the incidence-matrix products)
#   stores : stores of the saved streams skipped (forward: z1, z2, z3, geo; backward: dZ1, dD)
#   nomfma : the edge half's 64 x 64 products replaced by one LDS read each (what is left = everything that is not matrix math)
#   nonode : node half (forward) / node phase (backward) skipped
# then on the GPU box:  bash tools/ablation.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
names=${@:-silu seg stores nomfma nonode base}
mkdir -p "$root/gpurun_dbg"
for n in $names; do
  tmp=$(mktemp -d /tmp/abl.XXXXXX)
  cp "$root"/immunostruct_amd/csrc/*.hip "$root"/immunostruct_amd/csrc/*.h "$root"/immunostruct_amd/csrc/Makefile "$tmp"/
  flag="-DIS_ABL_$(echo $n | tr a-z A-Z)"
  [ "$n" = base ] && flag=""
  make -C "$tmp" -j8 EXTRA="$flag" > "$tmp/build.log" 2>&1 || { tail -20 "$tmp/build.log"; exit 1; }
  cp "$tmp/libimmunostruct_hip.so" "$root/gpurun_dbg/libimmunostruct_hip_abl_$n.so"
  rm -rf "$tmp"
  echo "gpurun_dbg/libimmunostruct_hip_abl_$n.so"
done
