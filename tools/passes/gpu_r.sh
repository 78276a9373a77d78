#!/bin/bash
# pass R: workgroup order rotated by the channel index on long rows (tile_rot): every XCD sees every part of the rows
mkdir -p gpurun_out
timeout 900 python -c "
import sys, os; sys.path.insert(0, os.getcwd())
import reevr_amd, pytest
assert reevr_amd.set_tuning('tile_rot', 1)
sys.exit(pytest.main(['tests/test_gpu_parity.py', '-m', 'gpu', '-q', '--timeout', '300', '-x', '-k', 'fuzz_block or lockstep or tiling or general_path or child_sets or guard']))
" 2>&1 | tail -3
run() {
  local label=$1; shift
  timeout 300 python bench.py --steps 8 --warmup 4 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'],2), v['frac']) for n, v in r['roofline_all'].items()}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': r['probe']['ok'], 'subsets': r['config']['subsets'], 'kernels(n,us,ms/step,frac)': k}))
" | tee -a gpurun_out/tune_r.jsonl
}
rm -f gpurun_out/tune_r.jsonl
run c2_sub1 --tune subsets=1
run c2_sub1_rot --tune subsets=1,tile_rot=1
run c2
run c2_rot --tune tile_rot=1
run c3_sub1 --config 3 --tune subsets=1
run c3_sub1_rot --config 3 --tune subsets=1,tile_rot=1
run c5 --config 5
run c5_rot --config 5 --tune tile_rot=1
