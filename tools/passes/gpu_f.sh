#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 420 2>&1 | tail -12 > gpurun_out/pytest_f.log
tail -4 gpurun_out/pytest_f.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_f.json 2> gpurun_out/bench_f.err
tail -c 400 gpurun_out/bench_f.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/bench_f.json').read().strip().splitlines()[-1])
print('value',r['value'],'ms',r['ms_per_step'],'probe',r['probe']['ok'],'subsets',r['config']['subsets'])
print('roofline',{k:r['roofline'][k] for k in ('kernel','frac','frac_per_launch','concurrency','alg_frac_reference_schedule','alg_equiv','traffic','bytes_per_launch')})
print('path',r['path_roofline']['executed_bytes_per_sample'],r['path_roofline']['frac_of_hbm_peak'])
for k,v in r['roofline_all'].items(): print(' ',k,v['launches_per_step'],v['avg_launch_ms'],v['ms_per_step'],v['concurrency'],v['frac'],v['frac_per_launch'])
print('ref', r['reference_schedule'])
print('stereo', r['stereo_block_sync']['modes'])
print('offline', r['stereo_offline_long_call']['value'], r['stereo_offline_fixed_partitions']['value'])
print('cpu', r['cpu_baseline']['value'], r['cpu_baseline']['all_cores']['value'])
for c in ('config1','config3','config5'):
    o=r[c]; print(c,o['value'],o['ms_per_step'],'exe_bps',o['executed_bytes_per_sample'],'exe_frac',o['frac_of_hbm_peak_executed_bytes'],'alg_equiv',o['alg_equiv'],'ref',o['reference_schedule']['value'],o['alg_frac_reference_schedule'],'probe',o['probe']['ok'],'cpu',o['cpu_baseline']['value'],o['cpu_baseline']['all_cores']['value'], 'tiles', o['tile_blocks'])
    for k,v in o['roofline_all'].items(): print('   ',k,v['launches_per_step'],v['avg_launch_ms'],v['ms_per_step'],v['concurrency'],v['frac'])
PY
bash tools/profile_bench.sh gpurun_out/prof_r3 > gpurun_out/prof_r3.log 2>&1
cat gpurun_out/prof_r3/kernel_union.txt | head -12
timeout 400 python tools/fence_fuzz.py 12 > gpurun_out/fence_fuzz.txt 2>&1; echo fence rc=$?; tail -3 gpurun_out/fence_fuzz.txt
