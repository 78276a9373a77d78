#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_h.json 2> gpurun_out/bench_h.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/bench_h.json').read().strip().splitlines()[-1])
print('value',r['value'],'ms',r['ms_per_step'],'probe',r['probe']['ok'],'subsets',r['config']['subsets'])
print('roofline',{k:r['roofline'][k] for k in ('kernel','frac','frac_per_launch','concurrency','alg_frac_reference_schedule','alg_equiv','traffic','bytes_per_launch','avg_launch_ms')})
print('path',r['path_roofline']['executed_bytes_per_sample'],r['path_roofline']['frac_of_hbm_peak'])
for k,v in r['roofline_all'].items(): print(' ',k,v['launches_per_step'],v['avg_launch_ms'],v['ms_per_step'],v['concurrency'],v['frac'],v['frac_per_launch'], v['traffic'])
for c in ('config1','config3','config5'):
    o=r[c]; print(c,o['value'],o['ms_per_step'],'exe_bps',o['executed_bytes_per_sample'],'exe_frac',o['frac_of_hbm_peak_executed_bytes'],'ref',o['reference_schedule']['value'],o['alg_frac_reference_schedule'],'probe',o['probe']['ok'],'cpu',o['cpu_baseline']['value'],o['cpu_baseline']['all_cores']['value'])
PY
bash tools/profile_bench.sh gpurun_out/prof_r3 > gpurun_out/prof_r3.log 2>&1
cat gpurun_out/prof_r3/kernel_union.txt | head -9
python - <<'PY'
import json
d=json.load(open('gpurun_out/prof_r3/traffic.json'))['config2']
for k,v in d['kernels'].items(): print(k, round(v['traffic_bytes']/1e6,1))
PY
