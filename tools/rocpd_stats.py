#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel stats table.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--top 40] > profiles/rNN_kernel_stats.txt
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    q = (f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc")
    rows = cur.execute(q).fetchall()
    total = sum(r[2] for r in rows)
    t0, t1 = cur.execute(f"select min(start), max(end) from {kd}").fetchone()
    print(f"# {args.db}: {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct kernels, "
          f"sum of kernel time {total / 1e6:.3f} ms, first->last dispatch span {(t1 - t0) / 1e6:.3f} ms")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
    for name, n, tot, mn, mx in rows[: args.top]:
        print(f"{n:7d} {tot / 1e6:10.3f} {tot / n / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100.0 * tot / total:6.2f}  {name[:150]}")
    if len(rows) > args.top:
        rest = rows[args.top:]
        print(f"{sum(r[1] for r in rest):7d} {sum(r[2] for r in rest) / 1e6:10.3f} {'':>10} {'':>9} {'':>9} "
              f"{100.0 * sum(r[2] for r in rest) / total:6.2f}  ({len(rest)} more kernels)")


if __name__ == "__main__":
    main()
