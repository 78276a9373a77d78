#!/bin/bash
mkdir -p gpurun_out
run() {
  local label=$1; shift
  timeout 300 python bench.py --steps 8 --warmup 4 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'],2), v['frac']) for n, v in r['roofline_all'].items()}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': r['probe']['ok'], 'tiles': r['config']['tile_blocks'], 'subsets': r['config']['subsets'], 'path_frac': r['path_roofline']['frac_of_hbm_peak'], 'kernels(n,us,ms/step,frac)': k}))
" | tee -a gpurun_out/tune_j.jsonl
}
rm -f gpurun_out/tune_j.jsonl
run c2_sub1_d8 --tune subsets=1,sweep_d=8
run c2_sub1_d8_lw2 --tune subsets=1,sweep_d=8,sweep_lw=2
run c3_sub1_k32_d8 --config 3 --tune subsets=1,sweep_d=8
run c3_sub1_k32_d4 --config 3 --tune subsets=1
run c1_sub1_k32_d8 --config 1 --tune subsets=1,sweep_d=8
run c2_d8 --tune sweep_d=8
run c3_default --config 3
run c3_d8 --config 3 --tune sweep_d=8
run c1_default --config 1
run c1_d8 --config 1 --tune sweep_d=8
