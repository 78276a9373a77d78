#!/bin/bash
# pass V: inverse transform without add-stream registers when there is no add stream; 8192 channels; side configs alone
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "kat or synth or bare or lockstep or general_path or one_big_call or single_stage or mixed" 2>&1 | tail -3
run() {
  local label=$1; shift
  timeout 400 python bench.py --steps 8 --warmup 4 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'],2), v['frac']) for n, v in (r.get('roofline_all') or {}).items()}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': (r.get('probe') or {}).get('ok'), 'subsets': r['config'].get('subsets'), 'kernels(n,us,ms/step,frac)': k}))
" | tee -a gpurun_out/tune_v.jsonl
}
rm -f gpurun_out/tune_v.jsonl
run c2_sub1 --tune subsets=1
run c2
run c2_8192 --channels 8192
run c2_8192_sub1 --channels 8192 --tune subsets=1
run c2_8192_sub4 --channels 8192 --tune subsets=4
