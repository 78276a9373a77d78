#!/bin/bash
# the profile part of gpu_k.sh alone (rocprofv3 kernel trace with the driver's step counts + PMC passes)
mkdir -p gpurun_out
bash tools/profile_bench.sh gpurun_out/prof_r3 > gpurun_out/prof_r3.log 2>&1
cat gpurun_out/prof_r3/kernel_union.txt | head -10
python - <<'PY'
import json
txt=[l for l in open('gpurun_out/prof_r3/bench_under_rocprof.json') if l.startswith('{')]
r=json.loads(txt[-1])
print(r['value'], {k:(round(v['avg_launch_ms']*1e3,1), v['concurrency']) for k,v in r['roofline_all'].items()})
PY
