#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "lockstep or large_head or guard_mode or block_synchronous_time_tiling" 2>&1 | tail -3
run() {
  local label=$1; shift
  timeout 300 python bench.py --steps 8 --warmup 4 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'],2), v['frac']) for n, v in r['roofline_all'].items()}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': r['probe']['ok'], 'subsets': r['config']['subsets'], 'path_frac': r['path_roofline']['frac_of_hbm_peak'], 'kernels(n,us,ms/step,frac)': k}))
" | tee -a gpurun_out/tune_m.jsonl
}
rm -f gpurun_out/tune_m.jsonl
run c2_sub1_nt --tune subsets=1
run c2_sub1_plain --tune subsets=1,patch_nt=0
run c2_nt
run c2_plain --tune patch_nt=0
run c2_nt_b
run c2_plain_b --tune patch_nt=0
