#!/bin/bash
# pass X: the plug-in's host loop (examples/host_block_loop) on the round-3 kernels: block, blocks, quad, idle us, persistent
mkdir -p gpurun_out
cd examples
for args in "512 3000 0 300 0" "512 3000 1 300 0" "512 3000 0 300 1" "512 3000 1 300 1" "512 3000 0 0 0" "512 3000 0 0 1" "480 3000 0 300 0" "256 3000 0 300 0"; do
  timeout 120 ./host_block_loop $args | tail -1
done | tee ../gpurun_out/hostloop_r3.txt
