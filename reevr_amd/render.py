"""Offline convolution-reverb renderer around the batched path:

    python -m reevr_amd.render --ir hall.wav --in dry.wav --out wet.wav [--wet 1.0 --dry 0.0]
                               [--block 512] [--tail] [--reverse --attack 0.0 --decay 1.0 --gain 1.0]
    python -m reevr_amd.render --raw --ir irs.wav --in dry.wav [more.wav ...] --out wet.wav [--block 4096] [--tail]

--raw is the MANY-CHANNEL BATCH form (BASELINE config 5: 64 parallel mono channels, 5 s IR, block 4096): the N channels
of the input file(s) are N independent mono convolvers, channel c convolved with channel c of the impulse file AS GIVEN
(one impulse channel: shared by all) -- no impulse stages, no wet bus, wet signal only. That is exactly N x
TwoStageFFTConvolver::init + process (libs/FFTConvolver/TwoStageFFTConvolver.cpp:87-233; the fan-out of
src/dsp/StereoConvolver.cpp:33-42 widened to N), served by ONE ConvolverSet(N) through the engine's long-call path.
Under `python -m torch.distributed.run --nproc-per-node G -m reevr_amd.render --raw ...` the channels are dealt to the
ranks (shard.units_for_rank, unit = channel: rank = channel mod world), every rank renders its own channels on its own
GPU and ONE gather per chunk (RCCL all_gather_into_tensor; gloo on a shared device) brings them to rank 0, which writes
the file.


The impulse is prepared on the device (reevr_amd.Impulse: auto gain, reverse, trim, gain, clip,
envelope -- the reference's Impulse::recalcImpulse), handed to the convolvers without a host round
trip, the whole file goes through ONE process() call per convolver pair (the long-call path of the
engine) and the wet-bus epilogue (true-stereo sum, dry/wet) runs on the device too. Channel logic is
the plug-in's (src/dsp/Impulse.cpp:160-196, StereoConvolver.cpp:22-42): a mono IR feeds both sides,
a stereo IR is (LL, RR), a four-channel IR is (LL, LR, RL, RR) = true stereo.
"""
from __future__ import annotations

import argparse
import time

import numpy as np

from . import Impulse, ConvolverSet
from .hotswap import wet_mix_device
from .wavio import read_wav, write_wav


def render(x: np.ndarray, sr: int, ir: np.ndarray, ir_sr: int, block: int = 512, wet: float = 1.0, dry: float = 0.0,
           width: float = 1.0, tail: bool = False, device: int = 0, **imp_params) -> np.ndarray:
    """x: (1 or 2, frames) dry signal; ir: (1, 2 or 4, n) impulse. Returns (2, frames[+tail]).
    The wet bus is the plug-in's (src/PluginProcessor.cpp:1840-1876): mid/side width with its
    1 / (1 + width) normalisation (width 1 = plain stereo at half level), then dry * dry + wet * wet."""
    import torch
    if ir_sr != sr:
        raise ValueError(f"impulse is {ir_sr} Hz, input {sr} Hz: resample first (JUCE-side step in the plug-in)")
    x = np.atleast_2d(np.asarray(x, np.float32))
    if x.shape[0] == 1:
        x = np.concatenate([x, x])
    x = x[:2]
    ir = np.atleast_2d(np.asarray(ir, np.float32))
    nir = ir.shape[0]
    # Impulse::load's channel mapping (Impulse.cpp:160-196)
    if nir >= 4:
        raw = [ir[0], ir[3], ir[1], ir[2]]          # file order LL, LR, RL, RR -> (LL, RR, LR, RL)
    elif nir == 2:
        raw = [ir[0], ir[1]]
    else:
        raw = [ir[0], ir[0]]
    imp = Impulse(device)
    imp.prepare(float(sr))
    imp.setRaw(*raw, trim_tail=True)                # Impulse::load drops trailing silence below 1e-3
    for k, v in imp_params.items():
        setattr(imp, k, v)
    imp.recalcImpulse()
    if tail:                                        # let the reverb ring out
        x = np.concatenate([x, np.zeros((2, imp.size), np.float32)], axis=1)
    frames = x.shape[1]
    head = 1
    while head < block:
        head *= 2
    tailb = max(8192, 2 * head)                     # StereoConvolver::prepare (StereoConvolver.cpp:8-20)
    dx = torch.from_numpy(np.ascontiguousarray(x)).cuda(device)
    main = ConvolverSet(2, device)
    assert main.init_impulse(head, tailb, imp, [0, 1], frames), main.last_error_string
    y = main.process_device(dx)                     # LL <- L, RR <- R
    cur = [y[0], y[1]]
    if imp.isQuad:
        cross = ConvolverSet(2, device)
        assert cross.init_impulse(head, tailb, imp, [2, 3], frames), cross.last_error_string
        z = cross.process_device(dx)                # LR <- L, RL <- R
        cur += [z[0], z[1]]
    out = wet_mix_device(cur, width=width, drygain=dry, wetgain=wet, dry=[dx[0], dx[1]], device=device)
    torch.cuda.synchronize(device)
    return torch.stack(out).cpu().numpy()


class BatchRenderer:
    """N independent mono convolvers with their own impulses as given (raw mode): one ConvolverSet(N), TwoStage geometry
    derived from the host block like StereoConvolver::prepare (StereoConvolver.cpp:8-20), device-resident long calls.
    bench.py's literal `--config 5` drives this object."""

    def __init__(self, irs, block: int, max_len: int, device: int = 0):
        self.n = len(irs)
        self.head = 1
        while self.head < block:
            self.head *= 2
        self.tail = max(8192, 2 * self.head)
        self.max_len = int(max_len)
        self.set = ConvolverSet(self.n, device)
        if not self.set.init(block, self.tail, [np.asarray(h, np.float32) for h in irs], max_len=self.max_len):
            raise RuntimeError("init failed: " + self.set.last_error_string)

    def process_device(self, d_in, d_out=None, sync: bool = True, order: bool = True):
        """(n, frames <= max_len) float32 CUDA tensors; consecutive calls continue the stream."""
        return self.set.process_device(d_in, d_out, sync=sync, order=order)

    def render(self, x: np.ndarray) -> np.ndarray:
        """host array (n, frames) -> (n, frames), in calls of at most max_len frames"""
        import torch
        x = np.ascontiguousarray(np.atleast_2d(np.asarray(x, np.float32)))
        assert x.shape[0] == self.n
        out = np.empty_like(x)
        dev = torch.device("cuda", self.set.device)
        for a in range(0, x.shape[1], self.max_len):
            b = min(a + self.max_len, x.shape[1])
            y = self.process_device(torch.from_numpy(np.ascontiguousarray(x[:, a:b])).to(dev))
            out[:, a:b] = y.cpu().numpy()
        return out

    def close(self):
        self.set.close()


def _dist_env():
    """(dist module or None, rank, world) -- a process group exists only under torch.distributed.run"""
    import os
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return None, 0, 1
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        same = os.environ.get("REEVR_BENCH_SAME_DEVICE") == "1"       # development: every rank on GPU 0, gloo
        lr = 0 if same else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        if same:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", lr))
    return dist, dist.get_rank(), dist.get_world_size()


def render_raw(x: np.ndarray, irs, block: int = 4096, tail: bool = False, device: int = 0, chunk: int = 1 << 22,
               dist=None, rank: int = 0, world: int = 1):
    """x: (N, frames) dry channels; irs: N impulses (any lengths) or ONE shared by all channels. Returns the (N, frames
    [+ longest impulse]) wet channels on rank 0 (None on the other ranks). Channels are dealt to the ranks with
    shard.units_for_rank; shards are padded with silent channels to equal size for the one gather per chunk."""
    import torch
    from . import shard
    x = np.atleast_2d(np.asarray(x, np.float32))
    N = x.shape[0]
    irs = [np.asarray(h, np.float32).reshape(-1) for h in irs]
    if len(irs) == 1:
        irs = irs * N
    if len(irs) != N:
        raise ValueError(f"{N} input channels but {len(irs)} impulse channels (need one per channel, or one for all)")
    if tail:
        x = np.concatenate([x, np.zeros((N, max(h.size for h in irs)), np.float32)], axis=1)
    frames = x.shape[1]
    mine = shard.units_for_rank(N, world, rank)
    per = -(-N // world)                                   # channels per rank incl. padding
    silent = np.zeros(1, np.float32)
    my_irs = [irs[c] for c in mine] + [silent] * (per - len(mine))
    my_x = np.zeros((per, frames), np.float32)
    if mine:
        my_x[:len(mine)] = x[mine]
    max_len = max(1, min(frames, chunk))
    r = BatchRenderer(my_irs, block, max_len, device)
    dev = torch.device("cuda", device)
    out = np.empty((N, frames), np.float32) if rank == 0 else None
    for a in range(0, frames, max_len):
        b = min(a + max_len, frames)
        y = r.process_device(torch.from_numpy(np.ascontiguousarray(my_x[:, a:b])).to(dev))
        g = shard.gather_batches(y, dist)                                  # (world, per, b - a); one rank under the launcher: RCCL too
        if rank == 0:
            g = g.cpu().numpy()
            for rr in range(world):
                for j, c in enumerate(shard.units_for_rank(N, world, rr)):
                    out[c, a:b] = g[rr, j]
    r.close()
    return out


def main_raw(a) -> int:
    dist, rank, world = _dist_env()
    import os
    device = a.device if dist is None else (0 if os.environ.get("REEVR_BENCH_SAME_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0")))
    xs, sr = [], None
    for path in a.inp:
        x, s = read_wav(path)
        if sr is not None and s != sr:
            raise ValueError(f"{path}: {s} Hz, the other inputs {sr} Hz")
        sr = s
        xs.append(x)
    frames = max(x.shape[1] for x in xs)
    x = np.concatenate([np.pad(x, ((0, 0), (0, frames - x.shape[1]))) for x in xs])     # channels of all files, zero-padded
    ir, ir_sr = read_wav(a.ir)
    if ir_sr != sr:
        raise ValueError(f"impulse is {ir_sr} Hz, input {sr} Hz: resample first (JUCE-side step in the plug-in)")
    t = time.perf_counter()
    y = render_raw(x, list(ir), block=a.block, tail=a.tail, device=device, dist=dist, rank=rank, world=world)
    dt = time.perf_counter() - t
    if rank == 0:
        write_wav(a.out, y, sr, float32=not a.pcm16)
        print(f"{a.out}: {y.shape[0]} channels x {y.shape[1] / sr:.2f} s through {ir.shape[0]} impulse(s) of {ir.shape[1] / sr:.2f} s "
              f"on {world} GPU(s) in {dt * 1e3:.0f} ms ({y.shape[0] * y.shape[1] / dt / 1e6:.0f} Msamples/s incl. upload, init, gather and download)")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ir", required=True)
    ap.add_argument("--in", dest="inp", required=True, nargs="+")
    ap.add_argument("--out", required=True)
    ap.add_argument("--raw", action="store_true", help="many-channel batch: channel c of the input(s) through channel c of the impulse file as given, wet only")
    ap.add_argument("--block", type=int, default=0, help="host block size the plug-in would run at (sets the partition sizes; default 512, --raw: 4096)")
    ap.add_argument("--wet", type=float, default=1.0)
    ap.add_argument("--dry", type=float, default=0.0)
    ap.add_argument("--width", type=float, default=1.0, help="mid/side width of the wet bus, normalised by 1 / (1 + width) like the plug-in")
    ap.add_argument("--tail", action="store_true", help="append the impulse length of silence so the reverb rings out")
    ap.add_argument("--pcm16", action="store_true", help="write 16-bit PCM instead of float32")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--reverse", action="store_true")
    ap.add_argument("--attack", type=float, default=0.0)
    ap.add_argument("--decay", type=float, default=1.0)
    ap.add_argument("--gain", type=float, default=1.0)
    ap.add_argument("--trim-left", type=float, default=0.0)
    ap.add_argument("--trim-right", type=float, default=0.0)
    a = ap.parse_args(argv)
    if a.raw:
        a.block = a.block or 4096
        return main_raw(a)
    a.block = a.block or 512
    if len(a.inp) != 1:
        ap.error("several input files need --raw")
    x, sr = read_wav(a.inp[0])
    ir, ir_sr = read_wav(a.ir)
    t = time.perf_counter()
    y = render(x, sr, ir, ir_sr, block=a.block, wet=a.wet, dry=a.dry, width=a.width, tail=a.tail, device=a.device, reverse=a.reverse,
               attack=a.attack, decay=a.decay, gain=a.gain, trimLeft=a.trim_left, trimRight=a.trim_right)
    dt = time.perf_counter() - t
    write_wav(a.out, y, sr, float32=not a.pcm16)
    print(f"{a.out}: {y.shape[1] / sr:.2f} s of stereo audio through a {ir.shape[1] / sr:.2f} s impulse in {dt * 1e3:.0f} ms "
          f"({y.shape[1] / sr / dt:.0f} x real time incl. upload, init and download)")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
