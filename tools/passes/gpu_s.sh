#!/bin/bash
# pass S: second-level sweep epilogue with all first-level row requests issued before the stores; tile_rot default on
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "fuzz_block or lockstep or tiling or general_path or child_sets or guard or two_level" 2>&1 | tail -3
run() {
  local label=$1; shift
  timeout 300 python bench.py --steps 8 --warmup 4 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'],2), v['frac']) for n, v in (r.get('roofline_all') or {}).items()}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': r['probe']['ok'], 'subsets': r['config'].get('subsets'), 'kernels(n,us,ms/step,frac)': k}))
" | tee -a gpurun_out/tune_s.jsonl
}
rm -f gpurun_out/tune_s.jsonl
run c2_sub1 --tune subsets=1
run c2
run c2_norot --tune tile_rot=0
run c3_sub1 --config 3 --tune subsets=1
run c3 --config 3
run c1_sub1 --config 1 --tune subsets=1
run c1 --config 1
run c5 --config 5
run c5_norot --config 5 --tune tile_rot=0
