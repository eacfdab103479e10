#!/usr/bin/env python
"""Per-kernel PMC totals from a rocprofv3 rocpd database: sums the counter over its instances per dispatch,
then averages over dispatches.   python tools/rocpd_pmc.py x_results.db [substring]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else "is"
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
t = lambda p: [x for x in tabs if x.startswith(p)][0]
pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
q = (f"select s.kernel_name, p.name, d.id, sum(e.value), d.end - d.start from {pe} e join {ip} p on e.pmc_id = p.id "
     f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by d.id, p.name")
agg = defaultdict(list)
for name, pmc, did, val, dur in cur.execute(q):
    if filt in name:
        agg[(name.split("(")[0][:70], pmc)].append((val, dur))
print(f"{'kernel':72s} {'counter':14s} {'launches':>8s} {'mean per launch':>16s} {'mean us':>9s}")
for (name, pmc), vals in sorted(agg.items()):
    print(f"{name:72s} {pmc:14s} {len(vals):8d} {sum(v for v, _ in vals) / len(vals):16.1f} {sum(d for _, d in vals) / len(vals) / 1e3:9.2f}")
