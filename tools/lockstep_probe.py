#!/usr/bin/env python3
"""Block-synchronous loop over C lock-step channels (SURVEY.md 8d's HBM-bound regime): one
process_device() call per 512-frame block for ALL channels of one set, C large enough that the IR
spectra + delay lines leave the 256 MiB Infinity Cache. Prints per-kernel HIP-event times too.

  python tools/lockstep_probe.py [channels ...]        e.g.  128 512 1024
env: PROBE_BLOCKS (default 256), PROBE_BG (1: tail job on the second stream), PROBE_TIMING (0/1)
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reevr_amd
from reevr_amd import KERNEL_NAMES, synth

SR, IR_LEN, BLK, TAIL = 48000, 480000, 512, 8192
ALG_BPS = 1497.0


def run(nch: int, nblk: int, bg: bool, timing: bool):
    base = [synth.synth_ir(IR_LEN, 2, inst=i) for i in range(4)]          # 8 distinct IRs, cycled: every
    irs = [base[(c // 2) % 4][c % 2] for c in range(nch)]                 # channel still owns its own spectra in HBM
    s = reevr_amd.ConvolverSet(nch, bg_stream=bg)
    t0 = time.perf_counter()
    assert s.init(BLK, TAIL, irs, max_len=BLK), s.last_error_string
    s.sync()
    init_s = time.perf_counter() - t0
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.rand((nch, nblk * BLK), device="cuda", generator=g) * 2 - 1)
    y = torch.empty_like(x)
    # warm-up: the tail delay line holds 57 tail blocks of history -- rows before time 0 are never fetched
    # (clamped to one cached row), so the steady state starts after 59 tail periods = 944 blocks
    for _ in range(-(-960 // nblk)):
        s.process_device_blocks(x, BLK)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.process_device_blocks(x, BLK, y)
    dt = time.perf_counter() - t0
    rate = nch * nblk * BLK / dt
    rec = dict(channels=nch, blocks=nblk, bg=bg, init_s=round(init_s, 2), us_per_block=round(dt / nblk * 1e6, 2),
               Msamples_s=round(rate / 1e6, 1), alg_GBs=round(rate * ALG_BPS / 1e9, 1), frac=round(rate * ALG_BPS / 8e12, 4))
    if timing:
        s.set_timing(True)
        s.kernel_time_reset()
        s.process_device_blocks(x[:, :64 * BLK].contiguous(), BLK, y[:, :64 * BLK].contiguous())
        k = {}
        for kid, name in enumerate(KERNEL_NAMES):
            n, ms = s.kernel_time(kid)
            if n:
                k[name] = dict(launches=n, avg_us=round(ms / n * 1e3, 2))
        rec["kernels"] = k
        s.set_timing(False)
    print(json.dumps(rec), flush=True)
    s.close()
    del x, y
    torch.cuda.empty_cache()


if __name__ == "__main__":
    chans = [int(a) for a in sys.argv[1:]] or [2, 16, 128, 512]
    nblk = int(os.environ.get("PROBE_BLOCKS", "256"))
    for c in chans:
        run(c, nblk, os.environ.get("PROBE_BG", "0") == "1", os.environ.get("PROBE_TIMING", "1") == "1")
