#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/guard_diag.py 226 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 > gpurun_out/guard_diag.txt 2>&1
tail -40 gpurun_out/guard_diag.txt
