mkdir -p gpurun_out/c14
for tree in gpurun_dbg/r02tree .; do tag=$( [ $tree = . ] && echo cur || echo r02 ); for w in iedb paired; do (cd $tree && python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/c14/${tag}_$w.json); done; done
python - <<'PY'
import json
for w in ("iedb","paired"):
    a=json.load(open(f'gpurun_out/c14/r02_{w}.json')); b=json.load(open(f'gpurun_out/c14/cur_{w}.json'))
    print(w, a['ms_per_step'], b['ms_per_step'])
    ka, kb = a['kernel_timers_us'], b['kernel_timers_us']
    for k in sorted(set(ka)|set(kb)):
        x, y = ka.get(k,[0,0]), kb.get(k,[0,0])
        if abs(y[1]*y[0]-x[1]*x[0])/10 > 1.0: print(f"   {k:34s} r02 {x[0]:4d} x {x[1]:7.2f}   cur {y[0]:4d} x {y[1]:7.2f}   per-step d {(y[1]*y[0]-x[1]*x[0])/10:+.1f}")
PY
