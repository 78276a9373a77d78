"""Repeat one parity case many times in one process and report mismatches (race hunting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reevr_amd
from tests import cases

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_stereo_10s_b512"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
case = cases.SYNTH_CASES[name]
irs = cases.make_ir(case["ir"])
x = cases.make_input(case, irs.shape[0])
ref = None
bad = 0
for i in range(reps):
    s = reevr_amd.ConvolverSet(irs.shape[0])
    assert s.init(case["head"], case["tail"], list(irs), max_len=case["frames"])
    out = s.process(x)
    s.clear()
    cut = case["frames"] // 3 + 7
    dx = torch.from_numpy(x).cuda()
    o1 = s.process_device(dx[:, :cut].contiguous())
    o2 = s.process_device(dx[:, cut:].contiguous())
    out2 = torch.cat([o1, o2], dim=1).cpu().numpy()
    if ref is None:
        ref = out.copy()
    for tag, o in (("host one call", out), ("device two calls", out2)):
        d = np.abs(o.astype(np.float64) - ref)
        if d.max() > 1e-4:
            bad += 1
            idx = np.argwhere(d > 1e-4)
            print(f"rep {i} {tag}: {idx.shape[0]} samples differ, channels {sorted(set(idx[:,0]))}, first {idx[0]}, last {idx[-1]}, max {d.max():.3e}")
    s.close()
    # churn the allocator like the test-suite does
    junk = [torch.empty(int(np.random.randint(1, 64)) * 1024 * 256, device="cuda") for _ in range(3)]
    del junk
print("reps", reps, "bad", bad)
