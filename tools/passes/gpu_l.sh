#!/bin/bash
mkdir -p gpurun_out
PMC_ARGS="--config 3 --head-log 8 --k1-head 16 --k1-tail 32 --channels 2048" bash tools/profile_bench.sh gpurun_out/prof_r3_c3 --config 3 > gpurun_out/prof_r3_c3.log 2>&1
head -12 gpurun_out/prof_r3_c3/kernel_union.txt
PMC_ARGS="--config 1 --head-log 9 --tail-log 99 --k1-head 32 --k1-tail 0 --channels 8192" bash tools/profile_bench.sh gpurun_out/prof_r3_c1 --config 1 > gpurun_out/prof_r3_c1.log 2>&1
head -6 gpurun_out/prof_r3_c1/kernel_union.txt
