#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer boundary (rvc_set_process on HOST buffers, one call per 512-frame block: staging into
pinned memory, H2D, kernels, D2H, copy out -- what a host that keeps its audio in host memory sees), BASELINE config 2's geometry
at several channel counts. bench.py's `value` is the device-resident rate; this is the number beside it for DESIGN.md.
   python tools/host_rate.py [channels ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reevr_amd
from reevr_amd import synth


def main():
    chans = [int(a) for a in sys.argv[1:]] or [2, 64, 1024, 4096]
    head, tail, ir_len = 512, 8192, 480000
    base = [synth.synth_ir(ir_len, 2, inst=i) for i in range(8)]
    for nch in chans:
        irs = [base[(c // 2) % 8][c % 2] for c in range(nch)]
        nblk = 64 if nch >= 1024 else 400
        x = np.stack([synth.synth_input(head * nblk, c % 16) for c in range(nch)])
        s = reevr_amd.ConvolverSet(nch)
        assert s.init(head, tail, irs, max_len=head), s.last_error_string
        s.process_host_blocks_timed(x[:, :head * 8], head)                      # warm
        t0 = time.perf_counter()
        _, us = s.process_host_blocks_timed(x, head)
        el = time.perf_counter() - t0
        s.check()
        us = np.sort(us)
        print(json.dumps({"channels": nch, "host_block": head, "blocks": nblk, "Msamples_s_pcie_inclusive": round(nch * head * nblk / el / 1e6, 1),
                          "call_us_median": round(float(us[len(us) // 2]), 1), "call_us_p99": round(float(us[int(len(us) * 0.99)]), 1),
                          "MB_per_call_each_way": round(nch * head * 4 / 1e6, 2)}), flush=True)
        s.close()


if __name__ == "__main__":
    main()
