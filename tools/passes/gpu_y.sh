#!/bin/bash
# pass Y: calls across one head-block boundary as two latency-path steps
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 420 -x 2>&1 | tail -4
cd examples
for args in "512 3000 0 300 0" "480 3000 0 300 0" "480 3000 1 300 0" "441 3000 0 300 0" "480 3000 0 0 0" "480 3000 0 300 1" "256 3000 0 300 0" "1024 3000 0 300 0"; do
  timeout 120 ./host_block_loop $args | tail -1
done | tee ../gpurun_out/hostloop_r3b.txt
