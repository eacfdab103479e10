#!/bin/bash
# usage (here, after `gpurun -- 'bash tools/round_profiles.sh TAG'`): bash tools/adopt_evidence.sh TAG rNN  -- copies the evidence set
# gpurun_out/TAG/ over the tracked profiles/rNN_* files (the RCCL line: last JSON line only; the traffic file gets the commit label
# and is checked against the tree's kernel-source digest)
s=gpurun_out/$1; r=$2
for w in iedb iedb_finetune paired stress; do cp $s/bench_$w.json profiles/${r}_bench_$w.json; done
grep -E '^\{' $s/bench_iedb_rccl1.json | tail -1 > profiles/${r}_bench_iedb_rccl1.json
cp $s/kernel_stats.txt profiles/${r}_graph_replay_kernel_stats.txt
cp $s/timeline.txt profiles/${r}_step_timeline.txt
cp $s/sweep.json profiles/${r}_sweep.json
for k in layer_bwd layer_fwd node_wgrad; do cp $s/sq_counters_$k.txt profiles/${r}_sq_counters_$k.txt; done
python - "$s" "$r" <<'PY'
import json, subprocess, sys, bench
s, r = sys.argv[1], sys.argv[2]
d = json.load(open(f"{s}/pmc_traffic.json"))
assert d["kernel_sources_sha256"] == bench.kernel_sources_sha256(), "the traffic file was measured on other kernel sources"
d["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
json.dump(d, open(f"profiles/{r}_pmc_traffic.json", "w"), indent=1)
print("adopted", s, "->", f"profiles/{r}_*", "traffic digest ok")
PY
