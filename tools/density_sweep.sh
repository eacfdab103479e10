#!/bin/bash
# usage (on the GPU box): bash tools/density_sweep.sh OUT.json   -- SURVEY 8(d) config 2: the real edge count is unknown, so the
# default bench line is taken at deg_extra = 1, 2, 5, 8 (E/N = 2, 3, 6, 9), regression stage, at two SYMMETRIC variants (every
# chain link and contact in both directions, as from_networkx lists the undirected residue graphs: deg_extra = 0 and 2, E/N = 2
# and 6 -- to be compared with the directed rows of equal E), plus the BCE finetune stage at deg_extra = 2; one JSON array of
# {deg_extra, symmetric, stage, edges_per_batch, value, ms_per_step, step_ms median, in-situ layer-kernel us}
out=$1
echo "[" > $out
first=1
for spec in "1 pretrain 0" "2 pretrain 0" "5 pretrain 0" "8 pretrain 0" "0 pretrain 1" "2 pretrain 1" "2 finetune 0"; do
  set -- $spec
  sym=""; [ "$3" = 1 ] && sym="--symmetric-edges"
  line=$(python bench.py --deg-extra $1 --stage $2 $sym --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-copy-ceiling 2>/dev/null | tail -1)
  [ $first = 1 ] || echo "," >> $out
  first=0
  echo "$line" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print(json.dumps(dict(deg_extra=$1, symmetric=bool($3), stage='$2', edges_per_batch=d['config']['edges_per_batch'], value=d['value'], ms_per_step=d['ms_per_step'],
                      step_ms_median=d['step_ms']['median'], insitu_us=r.get('insitu_us'), frac=r.get('frac'))))" >> $out
done
echo "]" >> $out
cat $out
