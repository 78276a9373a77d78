#!/bin/bash
mkdir -p gpurun_out
run() {
  local label=$1; shift
  timeout 300 python bench.py --steps 8 --warmup 4 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'],2), v['frac']) for n, v in r['roofline_all'].items() if n in ('fused_block','sweep_head')}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': r['probe']['ok'], 'subsets': r['config']['subsets'], 'kernels(n,us,ms/step,frac)': k}))
" | tee -a gpurun_out/tune_p.jsonl
}
rm -f gpurun_out/tune_p.jsonl
run c2_sub1 --tune subsets=1
run c2_sub1_occ3 --tune subsets=1,patch_nt=3
run c2 
run c2_occ3 --tune patch_nt=3
run c2_8192_sub1 --channels 8192 --tune subsets=1
run c2_8192_sub1_occ3 --channels 8192 --tune subsets=1,patch_nt=3
