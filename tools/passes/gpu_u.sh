#!/bin/bash
# pass U: zero-latency stage (32 partitions) with two levels (16 / 8) instead of one level of 8
mkdir -p gpurun_out
run() {
  local label=$1; shift
  timeout 300 python bench.py --steps 8 --warmup 4 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'],2), v['frac']) for n, v in (r.get('roofline_all') or {}).items()}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': r['probe']['ok'], 'subsets': r['config'].get('subsets'), 'tiles': r['config'].get('tile_rows'), 'path': r['path_roofline']['executed_bytes_per_sample'], 'kernels(n,us,ms/step,frac)': k}))
" | tee -a gpurun_out/tune_u.jsonl
}
rm -f gpurun_out/tune_u.jsonl
run c2_sub1 --tune subsets=1
run c2_sub1_two --tune subsets=1,two_level_min_p=20
run c2
run c2_two --tune two_level_min_p=20
run c5 --config 5
