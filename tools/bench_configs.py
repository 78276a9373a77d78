#!/usr/bin/env python3
"""Side measurements for DESIGN.md: the five BASELINE.json configurations, batched (one
process() call over the whole input, device-resident) and block-synchronous (one call per host
block, C loop), plus the PCIe-inclusive host-pointer call latency. Not the graded bench line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reevr_amd
from reevr_amd import synth

CONFIGS = [
    # name, channels, sr, ir_seconds, host block, uniform?, seconds of input
    ("cfg1 mono 1s IR b512 (FFTConvolver)", 1, 48000, 1.0, 512, True, 40),
    ("cfg2 stereo 10s IR b512", 2, 48000, 10.0, 512, False, 40),
    ("cfg3 stereo 30s IR @96k b256", 2, 96000, 30.0, 256, False, 40),
    ("cfg4 8 stereo instances 10s IR b512 (one GPU)", 16, 48000, 10.0, 512, False, 40),
    ("cfg5 64 mono channels 5s IR b4096", 64, 48000, 5.0, 4096, False, 20),
]

def main():
    out = []
    for name, nch, sr, irs_s, blk, uniform, secs in CONFIGS:
        ir_len = int(irs_s * sr)
        head = 1
        while head < blk: head *= 2
        tail = max(8192, 2 * head)
        frames = (secs * sr // blk) * blk
        irs = np.concatenate([synth.synth_ir(ir_len, 2, inst=i) for i in range((nch + 1) // 2)])[:nch]
        x = torch.from_numpy(np.stack([synth.synth_input(frames, c % 8) for c in range(nch)])).cuda()
        y = torch.empty_like(x)
        s = reevr_amd.ConvolverSet(nch)
        t0 = time.perf_counter()
        ok = s.init_uniform(blk, list(irs), max_len=frames) if uniform else s.init(blk, tail, list(irs), max_len=frames)
        s.sync(); init_ms = (time.perf_counter() - t0) * 1e3
        assert ok, s.last_error_string
        for _ in range(2): s.process_device(x, y)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps): s.process_device(x, y, sync=False, order=False)
        s.sync(); dt = (time.perf_counter() - t0) / reps
        batched = nch * frames / dt / 1e6
        # block-synchronous
        s2 = reevr_amd.ConvolverSet(nch, bg_stream=True)
        ok = s2.init_uniform(blk, list(irs), max_len=blk) if uniform else s2.init(blk, tail, list(irs), max_len=blk)
        nb = min(frames // blk, 2000)
        xs = x[:, :nb * blk].contiguous()
        s2.process_device_blocks(xs[:, :blk * 100].contiguous(), blk)
        t0 = time.perf_counter(); s2.process_device_blocks(xs, blk); dts = time.perf_counter() - t0
        stream = nch * nb * blk / dts / 1e6
        # host-pointer call (pinned staging + H2D + kernels + D2H + sync)
        xh = np.ascontiguousarray(xs[:, :blk * 300].cpu().numpy())
        for i in range(50): s2.process(xh[:, i * blk:(i + 1) * blk])
        lat = []
        for i in range(50, 300):
            t0 = time.perf_counter(); s2.process(xh[:, i * blk:(i + 1) * blk]); lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e6
        rec = dict(config=name, channels=nch, head=head, tail=0 if uniform else tail, partitions=[s.partitions(0), s.partitions(1)],
                   init_ms=round(init_ms, 1), batched_Msamples_s=round(batched, 1), batched_ms_per_call=round(dt * 1e3, 3),
                   block_sync_Msamples_s=round(stream, 1), block_sync_us_per_block=round(dts / nb * 1e6, 2),
                   host_call_us_median=round(float(np.median(lat)), 1), host_call_us_p99=round(float(np.percentile(lat, 99)), 1),
                   x_realtime_batched=round(batched * 1e6 / (sr * nch), 0))
        print(json.dumps(rec), flush=True)
        out.append(rec)
        s.close(); s2.close()
    return out

if __name__ == "__main__":
    main()
