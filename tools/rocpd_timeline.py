#!/usr/bin/env python
"""Per-dispatch timeline of the LAST train step in a rocprofv3 rocpd database (--kernel-trace).

A step is delimited by the first kernel of the step (default: the batch hand-over copy kernel).  Prints
start offset, duration, queue, gap to the previous dispatch end on the same queue, and the kernel name;
then the critical-path accounting of the main queue (busy time vs gaps).

    python tools/rocpd_timeline.py prof_results.db [--marker multi_copy] [--step -2]
"""
import argparse
import re
import sqlite3


def short(name):
    m = re.search(r"_ZN2is\d+([A-Za-z0-9_]+?)(?:I[LN]|E)", name)
    if m:
        return "is::" + m.group(1)
    if name.startswith("Cijk"):
        mt = re.search(r"MT\d+x\d+x\d+", name)
        return "hipblaslt " + name[:14] + " " + (mt.group(0) if mt else "")
    m = re.search(r"at6native\d*_?(?:GLOBAL__N_1)?\d*([A-Za-z_]+)", name)
    if m:
        tail = re.search(r"(FillFunctor|CUDAFunctor_add|MulFunctor|sum_functor|FusedAdam|threshold|clamp|direct_copy|CatArray|sigmoid|exp|MeanOps|tanh|log|DivFunctor|pow|sqrt)", name)
        return "at::" + m.group(1)[:28] + (" " + tail.group(1) if tail else "")
    return name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--marker", default="multi_copy")
    ap.add_argument("--step", type=int, default=-2, help="which marker occurrence starts the step (default: second to last)")
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    dcols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    qcol = "queue_id" if "queue_id" in dcols else ("stream_id" if "stream_id" in dcols else "0")
    rows = cur.execute(f"select d.start, d.end, d.{qcol}, s.{name_col} from {kd} d join {ks} s on d.kernel_id = s.id "
                       "order by d.start").fetchall()
    marks = [i for i, r in enumerate(rows) if args.marker in r[3]]
    if len(marks) < 2:
        raise SystemExit(f"marker {args.marker!r} found {len(marks)} times")
    lo = marks[args.step]
    hi = marks[args.step + 1] if args.step + 1 < 0 and args.step + 1 < len(marks) else len(rows)
    if args.step == -1:
        hi = len(rows)
    step = rows[lo:hi]
    t0 = step[0][0]
    last_end = {}
    busy = {}
    print(f"# step = dispatches {lo}..{hi - 1} of {len(rows)}; span {(max(r[1] for r in step) - t0) / 1e3:.1f} us")
    print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7} {'q':>3}  kernel")
    for s, e, q, name in step:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        busy[q] = busy.get(q, 0) + (e - s)
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} {q:>3}  {short(name)}")
    for q, b in busy.items():
        print(f"# queue {q}: busy {b / 1e3:.1f} us in {sum(1 for r in step if r[2] == q)} dispatches")
    # union of busy intervals over all queues
    iv = sorted((s, e) for s, e, _, _ in step)
    cov, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            cov += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    cov += ce - cs
    span = max(r[1] for r in step) - t0
    print(f"# GPU busy (union over queues) {cov / 1e3:.1f} us of {span / 1e3:.1f} us span; idle {(span - cov) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
