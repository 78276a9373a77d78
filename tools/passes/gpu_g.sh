#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tests/stress_fuzz.py 1000 120 > gpurun_out/stress_g.txt 2>&1; tail -4 gpurun_out/stress_g.txt
timeout 300 python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import torch, reevr_amd, bench
from reevr_amd import KERNEL_NAMES, synth
for sub in (-1, 1):
    reevr_amd.set_tuning("subsets", sub)
    o = bench.side_config(torch, reevr_amd, synth, KERNEL_NAMES, 3, 2048, 0, 6, 0.0)
    print(json.dumps({'label': 'c3_2048 subsets %d' % sub, 'value': o['value'], 'ms': o['ms_per_step'], 'ref': o['reference_schedule']['value'], 'probe': o['probe']['ok'], 'subsets': o['subsets'],
                  'kernels': {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'], 2), v['frac']) for n, v in o['roofline_all'].items()}}))
PY
