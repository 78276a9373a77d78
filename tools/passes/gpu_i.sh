#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "tiling or lockstep or guard or two_level or child or smallheads or small_heads or large_head or second_stream" 2>&1 | tail -6
run() {
  local label=$1; shift
  timeout 300 python bench.py --steps 8 --warmup 4 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'],2), v['frac']) for n, v in r['roofline_all'].items()}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': r['probe']['ok'], 'tiles': r['config']['tile_blocks'], 'subsets': r['config']['subsets'], 'path_frac': r['path_roofline']['frac_of_hbm_peak'], 'kernels(n,us,ms/step,frac)': k}))
" | tee -a gpurun_out/tune_i.jsonl
}
rm -f gpurun_out/tune_i.jsonl
run c2_sub1_k16 --tune subsets=1
run c2_sub1_k32 --tune subsets=1,k1=32
run c2_sub1_k8 --tune subsets=1,k1=8
run c2_sub1_k16_lw2 --tune subsets=1,sweep_lw=2
run c2_default
run c2_k32 --tune k1=32
run c2_bg --bg-stream 1
run c3_sub1_k16 --config 3 --tune subsets=1
run c3_sub1_k32 --config 3 --tune subsets=1,k1=32
run c3_k32 --config 3 --tune k1=32
run c1_sub1_k16 --config 1 --tune subsets=1
run c1_sub1_k32 --config 1 --tune subsets=1,k1=32
run c1_k32 --config 1 --tune k1=32
