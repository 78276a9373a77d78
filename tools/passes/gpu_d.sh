#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 420 2>&1 | tail -30 > gpurun_out/pytest_d.log
tail -6 gpurun_out/pytest_d.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err
tail -c 600 gpurun_out/bench_d.err
python - <<'PY'
import json
try:
    r=json.loads(open('gpurun_out/bench_d.json').read().strip().splitlines()[-1])
    print('value',r['value'],'ms',r['ms_per_step'],'probe',r['probe']['ok'],'subsets',r['config']['subsets'])
    print('roofline',{k:r['roofline'][k] for k in ('kernel','frac','frac_per_launch','concurrency','alg_frac_reference_schedule','alg_equiv')})
    print('path',r['path_roofline']['executed_bytes_per_sample'],r['path_roofline']['frac_of_hbm_peak'])
    for k,v in r['roofline_all'].items(): print(' ',k,v['launches_per_step'],v['avg_launch_ms'],v['ms_per_step'],v['concurrency'],v['frac'],v['frac_per_launch'])
    print('stereo', r['stereo_block_sync']['modes'])
    print('cpu', r['cpu_baseline']['value'], r['cpu_baseline']['all_cores']['value'])
    for c in ('config1','config3','config5'):
        if c in r:
            o=r[c]; print(c,o['value'],o['ms_per_step'],'exe_frac',o['frac_of_hbm_peak_executed_bytes'],'ref',o['reference_schedule']['value'],o['alg_frac_reference_schedule'],'probe',o['probe']['ok'],'cpu',o.get('cpu_baseline',{}).get('value'), 'sub', o['subsets'], 'init_ms', o['init_ms'], 'synth', o['synth_s'])
            for k,v in o['roofline_all'].items(): print('   ',k,v['launches_per_step'],v['avg_launch_ms'],v['ms_per_step'],v['frac'])
except Exception as e:
    print('bench parse failed',e)
PY
bash tools/profile_bench.sh gpurun_out/prof_r3 > gpurun_out/prof_r3.log 2>&1
tail -3 gpurun_out/prof_r3.log; cat gpurun_out/prof_r3/kernel_union.txt | head -20
