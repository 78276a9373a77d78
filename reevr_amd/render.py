"""Offline convolution-reverb renderer around the batched path:

    python -m reevr_amd.render --ir hall.wav --in dry.wav --out wet.wav [--wet 1.0 --dry 0.0]
                               [--block 512] [--tail] [--reverse --attack 0.0 --decay 1.0 --gain 1.0]

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


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ir", required=True)
    ap.add_argument("--in", dest="inp", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--block", type=int, default=512, help="host block size the plug-in would run at (sets the partition sizes)")
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
    x, sr = read_wav(a.inp)
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
