#!/usr/bin/env python
"""Merge the per-workload PMC summaries of tools/pmc_traffic.sh into ONE tracked file with the digest of the kernel sources they
were measured for (bench.py prints ``roofline.traffic: null`` + the reason when the tree's sources differ):

    python tools/assemble_traffic.py gpurun_out/TAG profiles/r03_pmc_traffic.json [commit label]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha256  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                                             cwd=ROOT).stdout.strip() or "?"
out = {"commit": label, "kernel_sources_sha256": kernel_sources_sha256(),
       "sources": ["immunostruct_amd/csrc/egnn_layer_fwd.hip", "immunostruct_amd/csrc/egnn_layer_bwd.hip", "immunostruct_amd/csrc/egnn_layer_bwd8.hip", "immunostruct_amd/csrc/common.h",
                   "immunostruct_amd/csrc/node16.h"],
       "method": "tools/pmc_traffic.sh per workload: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over "
                 "bench.py --eager --steps 3; bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction, MI355X_MICROARCH.md)"}
for wl, key in (("iedb", "iedb_B128_deg2"), ("paired", "paired_B128_deg2"), ("stress", "stress_B256")):
    path = os.path.join(src, f"pmc_{wl}.json")
    if os.path.isfile(path) and os.path.getsize(path) > 2:
        out[key] = json.load(open(path))
json.dump(out, open(dst, "w"), indent=1)
print(dst, {k: (v.get("egnn_layer_bwd_kernel", {}).get("bytes") if isinstance(v, dict) else None) for k, v in out.items() if k.endswith(("deg2", "B256"))})
