#!/bin/bash
# round-3 GPU pass A: the whole -m gpu suite, then the default bench line
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout 1500 python -m pytest tests -m gpu -q --timeout 420 -x --deselect tests/test_gpu_parity.py::test_lockstep_4096_channels_impulse_identity 2>&1 | tail -40 > gpurun_out/pytest_a.log
tail -5 gpurun_out/pytest_a.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
tail -c 1500 gpurun_out/bench_a.err
python - <<'PY'
import json
try:
    r=json.loads(open('gpurun_out/bench_a.json').read().strip().splitlines()[-1])
    print('value',r['value'],'ms',r['ms_per_step'],'probe',r['probe'])
    print('roofline',{k:r['roofline'][k] for k in ('kernel','frac','alg_frac_reference_schedule','alg_equiv')})
    for k,v in r['roofline_all'].items(): print(' ',k,v['launches_per_step'],v['avg_launch_ms'],v['frac'])
    for c in ('config1','config3','config5'):
        if c in r:
            o=r[c]; print(c,o['value'],o['ms_per_step'],'exe_frac',o['frac_of_hbm_peak_executed_bytes'],'ref',o['reference_schedule']['value'],o['alg_frac_reference_schedule'],'probe',o['probe'],'cpu',o.get('cpu_baseline',{}).get('value'))
            for k,v in o['roofline_all'].items(): print('   ',k,v['launches_per_step'],v['avg_launch_ms'],v['frac'])
except Exception as e:
    print('bench parse failed',e)
PY
