"""Phase breakdown of the big transform kernels from shader-clock stamps (build with -DRVC_PHASE_TIMES:
python tools/abl_build.py phase:-DRVC_PHASE_TIMES; REEVR_AMD_LIB=abl_libs/phase/libreevr_amd.so python tools/phase_times.py)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import reevr_amd
from reevr_amd import synth, _lib

nch, frames = 2, 1920000
irs = synth.synth_ir(480000, 2)
x = torch.from_numpy(np.stack([synth.synth_input(frames, c) for c in range(nch)])).cuda()
y = torch.empty_like(x)
s = reevr_amd.ConvolverSet(nch)
assert s.init(512, 8192, list(irs), max_len=frames)
for _ in range(3):
    s.process_device(x, y)
lib = _lib.lib()
buf = (C.c_ulonglong * (3 * 16 * 8))()
lib.rvc_debug_phase_times.argtypes = [C.c_void_p]
assert lib.rvc_debug_phase_times(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(3, 16, 8).astype(np.int64)
for k, name in ((0, "fwd"), (2, "inv")):
    d = np.diff(a[k][:, :5], axis=1)
    print(name, "cycles per phase (median over sampled workgroups):", np.median(d, axis=0).astype(int).tolist(),
          " start skew:", int(a[k][:, 0].max() - a[k][:, 0].min()))
d = np.diff(np.concatenate([a[0][:, 1:2], a[1][:, :8], a[0][:, 2:3]], axis=1), axis=1)
print("fwd core: [math0, xchg0, math1, xchg1, math2, xchg2, math3, xchg3, final radix-4]:", np.median(d, axis=0).astype(int).tolist())
