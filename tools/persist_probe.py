#!/usr/bin/env python3
"""Persistent block-synchronous mode (RVC_FLAG_PERSISTENT): quick parity + latency probe (development tool).
Run under `timeout`: a protocol bug shows as a hang."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reevr_amd
from reevr_amd import synth
from oracle import oracle_py as O

def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))

def oracle(irs, x, head, tail):
    out = []
    for c in range(x.shape[0]):
        o = O.TwoStageFFTConvolver("orc"); assert o.init(head, tail, irs[c]); out.append(o.process(x[c]))
    return np.stack(out)

def main():
    head, tail, nblk = 512, 8192, 300
    irs = synth.synth_ir(60000, 2, 1)
    x = np.stack([synth.synth_input(head * nblk, c) for c in range(2)])
    want = oracle(irs, x, head, tail)
    t0 = time.time()
    s = reevr_amd.ConvolverSet(2, persistent=True)
    assert s.init(head, tail, list(irs), max_len=4 * tail), s.last_error_string
    got = np.concatenate([s.process(x[:, i * head:(i + 1) * head]) for i in range(nblk)], axis=1)
    print("host blocks   rel rms", rel(got, want), "err", s.last_error, s.last_error_string, "%.2fs" % (time.time() - t0), flush=True)
    s.clear()
    dx = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    y = torch.cat([s.process_device(dx[:, i * head:(i + 1) * head].contiguous()) for i in range(nblk)], dim=1).cpu().numpy()
    print("device, one synchronous call per block rel rms", rel(y, want), s.last_error_string, flush=True)
    s.clear()
    y = s.process_device_blocks(dx, head).cpu().numpy()
    e = [rel(y[:, i * head:(i + 1) * head], want[:, i * head:(i + 1) * head]) for i in range(nblk)]
    print("device blocks (pipelined) rel rms", rel(y, want), "first bad block", next((i for i, v in enumerate(e) if v > 1e-5), None),
          "bad blocks", sum(v > 1e-5 for v in e), s.last_error_string, flush=True)
    s.clear()
    # ragged / mixed pattern: sub-block calls, a multi-block call, a long call, then blocks again
    sched, pos = [], 0
    for n in [512, 512, 100, 412, 512, 1536, 512, 512, 300, 212, 5 * 8192 + 17, 512 - 17] + [512] * 60:
        if pos + n > x.shape[1]: break
        sched.append(n); pos += n
    got = np.concatenate([s.process(x[:, a:a + n]) for a, n in zip(np.cumsum([0] + sched[:-1]), sched)], axis=1)
    print("mixed pattern rel rms", rel(got, want[:, :got.shape[1]]), s.last_error_string, flush=True)
    print("empty-command round trip: %.2f us (median of 2000)" % s._lib.rvc_debug_persist_rtt(s._h, 2000), flush=True)
    # latency: host-pointer per-block calls and the device-resident block loop
    s.clear()
    lat = []
    for i in range(nblk):
        t = time.perf_counter(); s.process(x[:, i * head:(i + 1) * head]); lat.append(time.perf_counter() - t)
    lat = np.array(lat[50:]) * 1e6
    print("host call us: median %.1f p10 %.1f p99 %.1f (python ctypes overhead included)" % (np.median(lat), np.percentile(lat, 10), np.percentile(lat, 99)), flush=True)
    big = torch.from_numpy(np.stack([synth.synth_input(head * 4000, c) for c in range(2)])).cuda()
    s.clear(); s.process_device_blocks(big[:, :head * 500].contiguous(), head)
    s.clear()
    t = time.perf_counter(); s.process_device_blocks(big, head); dt = time.perf_counter() - t
    print("device block loop: %.2f us per block (persistent)" % (dt / 4000 * 1e6), flush=True)
    s.close()
    s2 = reevr_amd.ConvolverSet(2)
    assert s2.init(head, tail, list(irs), max_len=head)
    s2.process_device_blocks(big[:, :head * 500].contiguous(), head)
    t = time.perf_counter(); s2.process_device_blocks(big, head); dt = time.perf_counter() - t
    print("device block loop: %.2f us per block (one launch per block)" % (dt / 4000 * 1e6), flush=True)
    s2.close()
    # parking: a short idle limit, then calls again
    os.environ["RVC_PERSIST_IDLE_MS"] = "30"
    s = reevr_amd.ConvolverSet(2, persistent=True)
    assert s.init(head, tail, list(irs), max_len=head)
    out = []
    for i in range(40):
        out.append(s.process(x[:, i * head:(i + 1) * head]))
        if i % 10 == 9: time.sleep(0.1)
    print("park/relaunch rel rms", rel(np.concatenate(out, axis=1), want[:, :40 * head]), s.last_error_string, flush=True)
    s.close()
    print("done", flush=True)

if __name__ == "__main__":
    main()
