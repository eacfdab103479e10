mkdir -p gpurun_out/c7
python -m pytest tests -m gpu -x -q > gpurun_out/c7/pytest.log 2>&1; tail -3 gpurun_out/c7/pytest.log
for v in flat auto flat auto; do IMMUNOSTRUCT_FWD_SHARES=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/c7/bench_$v.json 2>> gpurun_out/c7/bench.err; python -c "
import json;d=json.load(open('gpurun_out/c7/bench_$v.json'));r=d['roofline']['insitu_us'];print('$v',d['value'],d['ms_per_step'],d['step_ms']['median'],'e2e',d['e2e']['value'],'fwd slot',r['fwd']['slot']['mean'],'bwd slot',r['bwd']['slot']['mean'], 'kt', {k:v[1] for k,v in d['kernel_timers_us'].items() if 'comb' in k or 'attn' in k})"; done
for w in paired stress; do for v in flat auto; do IMMUNOSTRUCT_FWD_SHARES=$v python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c7/bench_${w}_$v.json 2>> gpurun_out/c7/bench.err; python -c "
import json;d=json.load(open('gpurun_out/c7/bench_${w}_$v.json'));print('$w $v',d['value'],d['ms_per_step'])"; done; done
timeout 600 python tools/dp_overlap_emulation.py --channels 16,32 --reserved 0,16 0 150 300 450 > gpurun_out/c7/dp_emulation.jsonl 2> gpurun_out/c7/dp_emulation.err; cat gpurun_out/c7/dp_emulation.jsonl; tail -2 gpurun_out/c7/dp_emulation.err | cut -c1-300
