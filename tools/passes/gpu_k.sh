#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 420 2>&1 | tail -5 > gpurun_out/pytest_k.log
tail -3 gpurun_out/pytest_k.log
bash tools/passes/gpu_h.sh
timeout 400 python tools/fence_fuzz.py 12 > gpurun_out/fence_fuzz.txt 2>&1; echo fence rc=$?; tail -2 gpurun_out/fence_fuzz.txt
