#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_o.json 2> gpurun_out/bench_o.err; echo rc=$?
python - <<'PY'
import json
r=json.loads(open('gpurun_out/bench_o.json').read().strip().splitlines()[-1])
print('value',r['value'],'ms',r['ms_per_step'],'probe',r['probe']['ok'])
print('roofline',{k:r['roofline'].get(k) for k in ('kernel','frac','frac_per_launch','frac_single_queue','concurrency','alg_frac_reference_schedule','alg_equiv')})
q=r['single_queue']; print('single_queue', q['value'], q['probe_ok'], q['roofline'])
for k,v in q['roofline_all'].items(): print('   ',k,v['launches_per_step'],v['avg_launch_ms'],v['frac'])
for c in ('config1','config3','config5'):
    o=r[c]; print(c,o['value'],o['frac_of_hbm_peak_executed_bytes'],o['alg_frac_reference_schedule'],o['probe']['ok'])
PY
