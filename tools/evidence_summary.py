#!/usr/bin/env python
"""python tools/evidence_summary.py gpurun_out/TAG -- one screen of what tools/round_profiles.sh wrote (for the docs)"""
import json
import sys

d = sys.argv[1]
for f in ("bench_iedb", "bench_iedb_finetune", "bench_paired", "bench_stress"):
    b = json.load(open(f"{d}/{f}.json"))
    r = b.get("roofline") or {}
    i = r.get("insitu_us") or {}
    c = b.get("cpu_baseline") or {}
    g = lambda k, w: ((i.get(k) or {}).get(w) or {}).get("mean")
    print(f, b["value"], b["ms_per_step"], "median", b["step_ms"]["median"], "frac", r.get("frac"), "ach", r.get("achieved"),
          "bwd", g("bwd", "slot"), g("bwd", "span"), "fwd", g("fwd", "slot"), g("fwd", "span"), "e2e", (b.get("e2e") or {}).get("value"))
    print("   cpu", c.get("value"), "1t", (c.get("one_thread") or {}).get("value"), "all", (c.get("all_cores") or {}).get("value"),
          "loader", (c.get("with_batch_construction") or {}).get("value"), "ceil", (b.get("hbm_copy_ceiling") or {}).get("value"),
          "fwd frac", (r.get("forward_kernel") or {}).get("frac_mfma"), (r.get("forward_kernel") or {}).get("tflops"),
          "gather", (r.get("gather_kernel") or {}).get("frac_of_measured"), "hbm", (r.get("hbm_view") or {}).get("traffic_gbs"),
          (r.get("hbm_view") or {}).get("traffic_frac_of_measured"))
line = json.loads([l for l in open(f"{d}/bench_iedb_rccl1.json") if l.lstrip().startswith("{")][-1])
print("rccl1", line["value"], line["ms_per_step"], line["config"]["dist_backend"], line["config"]["grad_allreduce"]["form"],
      line["config"]["grad_allreduce"]["tuned_ms"])
for r in json.load(open(f"{d}/sweep.json")):
    i = r["insitu_us"]
    print(r["deg_extra"], "sym" if r["symmetric"] else "dir", r["stage"], r["edges_per_batch"], r["value"], r["ms_per_step"],
          "fwd", i["fwd"]["slot"]["mean"], "bwd", i["bwd"]["slot"]["mean"])
