#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer boundary (rvc_set_process on HOST buffers, one call per 512-frame block: staging into
pinned memory, H2D, kernels, D2H, copy out -- what a host that keeps its audio in host memory sees), BASELINE config 2's geometry
at several channel counts; `in_place`: the caller produces / consumes its audio in the set's own staging rows
(rvc_set_host_buffers: no staging copy). bench.py's `value` is the device-resident rate; this is the number beside it.
   python tools/host_rate.py [channels ...] [--knobs k=v,...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reevr_amd
from reevr_amd import synth


def measure(nch, head=512, tail=8192, ir_len=480000, tune=None, base=None, nblk=None):
    base = base or [synth.synth_ir(ir_len, 2, inst=i) for i in range(8)]
    irs = [base[(c // 2) % 8][c % 2] for c in range(nch)]
    nblk = nblk or (96 if nch >= 1024 else 400)
    x = np.stack([synth.synth_input(head * nblk, c % 16) for c in range(nch)])
    s = reevr_amd.ConvolverSet(nch, tune=tune)
    assert s.init(head, tail, irs, max_len=head), s.last_error_string
    s.process_host_blocks_timed(x[:, :head * 8], head)                      # warm
    t0 = time.perf_counter()
    y, us = s.process_host_blocks_timed(x, head)
    el = time.perf_counter() - t0
    s.check()
    us = np.sort(us)
    rec = {"channels": nch, "host_block": head, "blocks": nblk, "Msamples_s_pcie_inclusive": round(nch * head * nblk / el / 1e6, 1),
           "call_us_median": round(float(us[len(us) // 2]), 1), "call_us_p99": round(float(us[int(len(us) * 0.99)]), 1),
           "MB_per_call_each_way": round(nch * head * 4 / 1e6, 2), "subsets": s.subsets, "tune": tune or {}}
    # in place: the caller's audio lives in the staging rows (here: filled from x per block, which a real producer would not need)
    s.clear()
    ins, outs = s.host_buffers()
    for c in range(nch):
        ins[c][:head] = x[c, :head]
    t_call = 0.0
    for b in range(nblk):
        t0 = time.perf_counter()
        s.process_in_place(head)
        t_call += time.perf_counter() - t0
    s.check()
    # (channels c and c % 16 share IR and input; they may sit in different phase groups / child sets: same samples to the last bits)
    ref_rms = float(np.sqrt(np.mean(outs[0][:head].astype(np.float64) ** 2))) + 1e-30
    same = all(float(np.sqrt(np.mean((outs[c][:head].astype(np.float64) - outs[c % 16][:head]) ** 2))) <= 1e-5 * ref_rms
               for c in range(16, nch, max(1, nch // 48)))
    rec["in_place_Msamples_s"] = round(nch * head * nblk / t_call / 1e6, 1)
    rec["in_place_call_us_mean"] = round(t_call / nblk * 1e6, 1)
    rec["in_place_consistent"] = bool(same)
    s.close()
    return rec


def main():
    args = [a for a in sys.argv[1:]]
    knobs = None
    if "--knobs" in args:
        i = args.index("--knobs")
        knobs = {k: int(v) for k, v in (kv.split("=") for kv in args[i + 1].split(","))}
        del args[i:i + 2]
    chans = [int(a) for a in args] or [2, 64, 1024, 4096]
    base = [synth.synth_ir(480000, 2, inst=i) for i in range(8)]
    for nch in chans:
        print(json.dumps(measure(nch, tune=knobs, base=base)), flush=True)


if __name__ == "__main__":
    main()
