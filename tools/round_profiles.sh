#!/bin/bash
# usage (on the GPU box): bash tools/round_profiles.sh TAG [COMMIT]   -- the evidence set of a round, written to gpurun_out/TAG/
# (COMMIT: the label stored in the traffic file -- the box has no .git: pass $(git rev-parse --short HEAD) from the build container):
#   bench_{iedb,paired,stress}.json  the JSON lines of python bench.py [--workload W]
#   kernel_stats.txt / timeline.txt  rocprofv3 --kernel-trace of a short default bench (per-kernel table, last step's timeline)
#   pmc_{iedb,paired,stress}.json    HBM traffic per launch (two PMC passes per workload, tools/pmc_traffic.sh)
#   pmc_traffic.json                 the three merged + the digest of the kernel sources (-> profiles/rNN_pmc_traffic.json)
#   bench_iedb_finetune.json, sweep.json   config 2's BCE stage; deg_extra = 1 / 2 / 5 / 8 (-> profiles/rNN_sweep.json)
tag=$1
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
# HBM traffic first: bench.py reports roofline.traffic only from a file whose kernel-source digest matches the tree, so the
# file of THIS build is put where bench.py looks for it (on the box; copy $out/pmc_traffic.json over the tracked one afterwards)
for wl in iedb paired stress; do
  bash tools/pmc_traffic.sh $wl $out/pmc_$wl.json
done
python tools/assemble_traffic.py $out $out/pmc_traffic.json ${2:-unlabelled}
cp $out/pmc_traffic.json $(python -c "import bench; print(bench.TRAFFIC_FILE)")
for wl in iedb paired stress; do
  python bench.py --workload $wl --steps 30 --warmup 5 > $out/bench_$wl.json 2> $out/bench_$wl.err
  cut -c1-230 $out/bench_$wl.json
done
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o rr -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-e2e --no-host-read > $out/prof.log 2> $out/prof.err
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_stats.py $db > $out/kernel_stats.txt 2>> $out/prof.err
python tools/rocpd_timeline.py $db > $out/timeline.txt 2>> $out/prof.err
# the data-parallel step under a one-rank RCCL group (the nccl backend for real on the one GPU: VERDICT r03 item 1), every form
# captured and timed (IMMUNOSTRUCT_DP_ONE_GRAPH=auto; the default since round 6 is 0: the multi-graph forms only)
IMMUNOSTRUCT_DP_ONE_GRAPH=auto IMMUNOSTRUCT_FORCE_COLLECTIVE=1 MASTER_PORT=29591 python bench.py --force-pack --steps 30 --warmup 5 --no-cpu-baseline --no-e2e > $out/bench_iedb_rccl1.json 2> $out/bench_iedb_rccl1.err
cut -c1-200 $out/bench_iedb_rccl1.json
# SQ counters of the two layer kernels and the node weight-gradient launch (instruction mix, LDS conflicts, waits)
bash tools/pmc_passes.sh $out/sq_counters_layer_bwd.txt egnn_layer_bwd
bash tools/pmc_passes.sh $out/sq_counters_layer_fwd.txt egnn_layer_fwd_kernel
bash tools/pmc_passes.sh $out/sq_counters_node_wgrad.txt egnn_node_wgrad16
# config 2's second stage and the edge-density sweep
python bench.py --stage finetune --steps 30 --warmup 5 > $out/bench_iedb_finetune.json 2> $out/bench_iedb_finetune.err
bash tools/density_sweep.sh $out/sweep.json > /dev/null
ls -la $out
