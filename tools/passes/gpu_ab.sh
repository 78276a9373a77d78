#!/bin/bash
# pass AB: first-level tail sweep variants once more on the final build (8 B per lane / deeper queue)
mkdir -p gpurun_out
run() {
  local label=$1; shift
  timeout 400 python bench.py --steps 8 --warmup 4 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'],2), v['frac']) for n, v in (r.get('roofline_all') or {}).items()}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': (r.get('probe') or {}).get('ok'), 'kernels(n,us,ms/step,frac)': k}))
" | tee -a gpurun_out/tune_ab.jsonl
}
rm -f gpurun_out/tune_ab.jsonl
run c2
run c2_lw2 --tune sweep_lw=2
run c2_d8 --tune sweep_d=8
run c2_lw2_d8 --tune sweep_lw=2,sweep_d=8
run c2_b
