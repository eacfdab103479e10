#!/usr/bin/env python
"""HBM traffic per launch of the HIP kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace
domain besides --kernel-trace) -> one JSON object per workload for profiles/rNN_pmc_traffic.json.

    python tools/pmc_traffic.py fetch_results.db write_results.db > entry.json

Counter unit = KiB.  gfx950 correction per MI355X_MICROARCH.md (HBM section): FETCH_SIZE under-reports coalesced reads by
exactly 2x -> bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (calibrated in round 1 on kernels with known byte counts).
Kernels are keyed by their demangled name up to the argument list, so template instantiations stay apart; the plain
"egnn_layer_bwd_kernel" / "egnn_layer_fwd_kernel" entries are the most frequent instantiation (the full launches)."""
import json
import re
import sqlite3
import subprocess
import sys
from collections import defaultdict


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, d.id, sum(e.value), d.end - d.start from {pe} e join {ip} p on e.pmc_id = p.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id where p.name = ? group by d.id")
    agg = defaultdict(list)
    for name, did, val, dur in cur.execute(q, (counter,)):
        agg[name].append((val, dur))
    return agg


def short(name):
    if name.startswith("_Z"):
        name = subprocess.run(["c++filt", name.replace(".kd", "")], capture_output=True, text=True).stdout.strip() or name
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name).replace("is::", "")


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for name in sorted(set(fetch) & set(write)):
        s = short(name)
        if not ("egnn" in s or "gather" in s or "reduce_partials" in s or "stack_prologue" in s or "attn" in s):
            continue
        f = sum(v for v, _ in fetch[name]) / len(fetch[name])
        w = sum(v for v, _ in write[name]) / len(write[name])
        us = sum(d for _, d in fetch[name]) / len(fetch[name]) / 1e3
        out[s] = dict(fetch_kib=round(f, 1), write_kib=round(w, 1), bytes=int((2 * f + w) * 1024), launches_sampled=len(fetch[name]),
                      mean_us_under_pmc=round(us, 2))
    # (the backward layer launch is egnn_layer_bwd_kernel -- two 256-thread workgroups per CU -- or, since round 5, its paired
    #  512-thread form egnn_layer_bwd8_kernel: whichever the workload launched most is "the backward layer kernel")
    for base, names in (("egnn_layer_bwd_kernel", ("egnn_layer_bwd_kernel", "egnn_layer_bwd8_kernel")),
                        ("egnn_layer_fwd_kernel", ("egnn_layer_fwd_kernel",))):
        cands = [(v["launches_sampled"], k) for k, v in out.items() if any(k.startswith(n + "<") for n in names)]
        if cands:
            out[base] = dict(out[max(cands)[1]], instantiation=max(cands)[1])
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
