#!/bin/bash
mkdir -p gpurun_out
run() {
  local label=$1; shift
  timeout 400 python bench.py --steps 8 --warmup 4 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'],2), v['frac']) for n, v in (r.get('roofline_all') or {}).items()}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': (r.get('probe') or {}).get('ok'), 'subsets': r['config'].get('subsets'), 'kernels(n,us,ms/step,frac)': k}))
" | tee -a gpurun_out/tune_z2.jsonl
}
rm -f gpurun_out/tune_z2.jsonl
run c2_sub1 --tune subsets=1
run c2_sub1_invloop --tune subsets=1,fft_loop=1
run c2_sub1_noloop --tune subsets=1,fft_loop=0
run c2
run c2_invloop --tune fft_loop=1
