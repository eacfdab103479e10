mkdir -p gpurun_out/c13
for rep in 1 2; do
for tree in gpurun_dbg/r02tree .; do for w in iedb paired; do (cd $tree && python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > /tmp/b.json; python -c "
import json;d=json.load(open('/tmp/b.json'));k=d['kernel_timers_us'];print('$tree $w',d['value'],d['ms_per_step'],d['step_ms']['median'],'bwd',k['egnn_layer_bwd'][1],'fwd',k['egnn_layer_fwd'][1])"); done; done; done
