#!/bin/bash
mkdir -p gpurun_out
run() {
  local label=$1; shift
  timeout 300 python bench.py --steps 8 --warmup 2 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'],2), v['frac']) for n, v in r['roofline_all'].items()}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': r['probe']['ok'], 'tiles': r['config']['tile_blocks'], 'subsets': r['config']['subsets'], 'path_frac': r['path_roofline']['frac_of_hbm_peak'], 'kernels(n,us,ms/step,frac)': k}))
" | tee -a gpurun_out/tune_e.jsonl
}
rm -f gpurun_out/tune_e.jsonl
run c2_default
run c2_stagger --tune fft_stagger=1
run c2_noloop_stagger --tune fft_loop=0,fft_stagger=1
run c2_sub1 --tune subsets=1
run c2_sub1_stagger --tune subsets=1,fft_stagger=1
run c2_sub1_noloop_stagger --tune subsets=1,fft_loop=0,fft_stagger=1

timeout 300 python - <<'PY' | tee -a gpurun_out/tune_e.jsonl
import json, sys, os
sys.path.insert(0, os.getcwd())
import torch, reevr_amd, bench
from reevr_amd import KERNEL_NAMES, synth
o = bench.side_config(torch, reevr_amd, synth, KERNEL_NAMES, 5, 4096, 0, 6, 0.0)
print(json.dumps({'label': 'c5_lockstep', 'value': o['value'], 'ms': o['ms_per_step'], 'ref': o['reference_schedule']['value'], 'probe': o['probe']['ok'],
                  'kernels': {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'], 2), v['frac']) for n, v in o['roofline_all'].items()}}))
PY
