#!/usr/bin/env python
"""Memory copies recorded beside the kernels in a rocprofv3 rocpd database (--kernel-trace --memory-copy-trace): count, sizes
and where the copies of the LAST step fall between its kernels (a copy node inside the captured step is invisible in the
kernel-only timeline and shows up there as an unexplained gap).

    python tools/rocpd_copies.py prof_results.db
"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
mc = [t for t in tabs if t.startswith("rocpd_memory_copy")]
print("tables:", [t for t in tabs if "copy" in t or "memory" in t])
if not mc:
    raise SystemExit("no memory-copy table")
cols = [r[1] for r in cur.execute(f"pragma table_info({mc[0]})")]
print("columns:", cols)
rows = cur.execute(f"select * from {mc[0]} order by start").fetchall()
print(len(rows), "copies")
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
k = cur.execute(f"select min(start), max(end) from {kd}").fetchone()
i_s, i_e, i_sz = cols.index("start"), cols.index("end"), cols.index("size") if "size" in cols else None
last = [r for r in rows if r[i_s] > k[1] - 3_000_000]      # the last 3 ms of kernel activity
for r in last[-40:]:
    print(f"  t-{(k[1] - r[i_s]) / 1e3:9.1f} us  dur {(r[i_e] - r[i_s]) / 1e3:7.1f} us  size {r[i_sz] if i_sz is not None else '?'}  {[r[cols.index(c)] for c in cols if c in ('name', 'kind', 'src_agent_id', 'dst_agent_id')]}")
