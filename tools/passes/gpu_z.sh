#!/bin/bash
# pass Z: rvc_kernels.hip without the SLP vectoriser (scalar FMAs instead of v_pk_* + register shuffles)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 420 -x 2>&1 | tail -4
run() {
  local label=$1; shift
  timeout 400 python bench.py --steps 8 --warmup 4 --side 0 --cpu-seconds 0 --distinct 64 "$@" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        k = {n: (round(v['launches_per_step'], 1), round(v['avg_launch_ms'] * 1e3, 1), round(v['ms_per_step'],2), v['frac']) for n, v in (r.get('roofline_all') or {}).items()}
        print(json.dumps({'label': '$label', 'value': r['value'], 'ms': r['ms_per_step'], 'probe': (r.get('probe') or {}).get('ok'), 'subsets': r['config'].get('subsets'), 'kernels(n,us,ms/step,frac)': k}))
" | tee -a gpurun_out/tune_z.jsonl
}
rm -f gpurun_out/tune_z.jsonl
run c2_sub1 --tune subsets=1
run c2
run c2_b
run c3 --config 3
run c1 --config 1
timeout 200 python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, reevr_amd
from reevr_amd import synth
irs = list(synth.synth_ir(480000, 2, 0))
xs = torch.from_numpy(np.stack([synth.synth_input(512 * 3000, c) for c in range(2)])).cuda()
for kw in (dict(), dict(fft_f32=True), dict(bg_stream=True)):
    s = reevr_amd.ConvolverSet(2, **kw); assert s.init(512, 8192, irs, max_len=512)
    s.process_device_blocks(xs[:, :512 * 200].contiguous(), 512)
    t = time.perf_counter(); s.process_device_blocks(xs, 512); dt = time.perf_counter() - t
    print('stereo pair', kw, round(dt / 3000 * 1e6, 2), 'us per block')
    s.close()
PY
