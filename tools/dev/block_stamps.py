#!/usr/bin/env python3
"""Where the time of ONE per-block launch goes (1024 lock-step channels, head block 512, time-tiled): per-workgroup
timestamps written by a development build of the library.
  python tools/abl_build.py stamps:-DRVC_BLOCK_STAMPS
  REEVR_AMD_LIB=abl_libs/stamps/libreevr_amd.so python tools/block_stamps.py [channels] [blocks after the tile start]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import reevr_amd
from reevr_amd import _lib, synth

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 5
BLK, TAIL, IR_LEN = 512, 8192, 480000
base = [synth.synth_ir(IR_LEN, 2, inst=i) for i in range(4)]
irs = [base[(c // 2) % 4][c % 2] for c in range(nch)]
s = reevr_amd.ConvolverSet(nch)
assert s.init(BLK, TAIL, irs, max_len=BLK), s.last_error_string
nblk = 256
x = torch.rand((nch, nblk * BLK), device="cuda") * 2 - 1
for _ in range(4):
    s.process_device_blocks(x, BLK)
s.process_device_blocks(x[:, :(8 + extra) * BLK].contiguous(), BLK)     # ends `extra` blocks into a sweep tile
s.sync()
lib = _lib.lib()
fn = lib.rvc_debug_block_stamps
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * (nch * 16))()
assert fn(buf, nch * 16)
t = np.frombuffer(buf, dtype=np.uint64).reshape(nch, 16).astype(np.int64)
t0 = min(t[:, 4].min(), t[:, 0][t[:, 0] > 0].min() if (t[:, 0] > 0).any() else t[:, 4].min())
us = lambda a: (a - t0) / 100.0          # wall_clock64: 100 MHz
names = {4: "audio wave starts", 5: "tables + IR rows + accumulator + previous spectrum arrived", 6: "+ tail stream", 8: "samples arrived, fold done",
         9: "forward transform done", 10: "split + MAC + inverse split done", 11: "inverse transform done", 12: "output stores acknowledged",
         0: "patch wave starts", 1: "patch wave: stores acknowledged"}
print(f"{nch} channels, block {extra} of its tile; microseconds after the first wave of the launch started")
for i in (4, 5, 6, 8, 9, 10, 11, 12, 0, 1):
    v = us(t[:, i])
    print(f"  {names[i]:62s} min {v.min():7.2f}  median {np.median(v):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")
d = (t[:, 12] - t[:, 4]) / 100.0
print(f"  audio wave lifetime: median {np.median(d):.2f} us, max {d.max():.2f}; patch wave lifetime: median {np.median((t[:, 1] - t[:, 0]) / 100.0):.2f}")
