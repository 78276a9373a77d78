#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "guard or large_head or child_sets or two_level or row_looping" 2>&1 | tail -8
run() {  # label, args...
  local label=$1; shift
  timeout 300 python bench.py --steps 6 --warmup 2 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), v['frac']) for n, v in r['roofline_all'].items()}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': r['probe']['ok'], 'tiles': r['config']['tile_blocks'], 'subsets': r['config']['subsets'], 'kernels_us': k}))
" | tee -a gpurun_out/tune_c.jsonl
}
rm -f gpurun_out/tune_c.jsonl
run c2_default
run c2_k8 --tune k1=8
run c2_k32 --tune k1=32
run c2_k16_lw4 --tune k1=16,sweep_lw=4
run c2_k16_split --tune k1=16,sweep_split=1
run c2_fftloop0 --tune fft_loop=0
run c2_sub2 --tune subsets=2
run c2_sub4 --tune subsets=4
run c2_8192 --channels 8192
run c2_8192_sub2 --channels 8192 --tune subsets=2
run c2_8192_sub4 --channels 8192 --tune subsets=4
run c3_1024 --config 3
run c3_1024_k32 --config 3 --tune k1=32
run c3_1024_k8 --config 3 --tune k1=8
run c3_2048 --config 3 --channels 2048
run c3_2048_sub2 --config 3 --channels 2048 --tune subsets=2
run c1_default --config 1
run c1_k32 --config 1 --tune k1=32
