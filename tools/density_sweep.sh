#!/bin/bash
# usage (on the GPU box): bash tools/density_sweep.sh OUT.json   -- SURVEY 8(d) config 2: the real edge count is unknown, so the
# default bench line is taken at deg_extra = 1, 2, 5, 8 (E/N = 2, 3, 6, 9), regression stage, plus the BCE finetune stage at
# deg_extra = 2; one JSON array of {deg_extra, stage, edges_per_batch, value, ms_per_step, step_ms median, in-situ layer-kernel us}
out=$1
echo "[" > $out
first=1
for spec in "1 pretrain" "2 pretrain" "5 pretrain" "8 pretrain" "2 finetune"; do
  set -- $spec
  line=$(python bench.py --deg-extra $1 --stage $2 --steps 30 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1)
  [ $first = 1 ] || echo "," >> $out
  first=0
  echo "$line" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print(json.dumps(dict(deg_extra=$1, stage='$2', edges_per_batch=d['config']['edges_per_batch'], value=d['value'], ms_per_step=d['ms_per_step'],
                      step_ms_median=d['step_ms']['median'], insitu_us=r.get('insitu_us'), frac=r.get('frac'))))" >> $out
done
echo "]" >> $out
cat $out
