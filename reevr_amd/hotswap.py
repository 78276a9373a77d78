"""IR hot-swap around two StereoConvolvers -- SURVEY.md 8(f) row f-2 (and the wet-bus sum of f-3).

Host-side mirror of the convolver section of the reference's processBlock
(src/PluginProcessor.cpp:1655-1756 warm-up ring + load state machine, :1793-1838 process,
crossfade, swap, wet sum; constants src/Globals.h:7-9), without JUCE:

  kIdle -> request_impulse() -> kLoading (worker thread: loadConvolver.loadImpulse)
        -> kReady  -> warm-up: the last 0.25 s of send signal replayed through the new convolver
                      (force2Chans) -- here ONE multi-block process() call instead of the
                      reference's loop of block-sized calls (same result: call-pattern independent)
        -> kFading -> CONV_XFADE ms linear crossfade old/new, then swap -> kIdle

The send-path IIR filters the reference applies to the replayed audio (irlowcut / irhighcut,
:1702-1745) are outside the convolution path; they are identity at their default settings and the
caller may pass an already filtered signal.
"""
from __future__ import annotations

import math
import threading

import numpy as np

CONV_XFADE_MS = 50            # src/Globals.h:7
K_IDLE, K_LOADING, K_READY, K_FADING = range(4)


class HotSwapStereoConvolver:
    def __init__(self, make_stereo_convolver, threaded: bool = True):
        """make_stereo_convolver() -> object with the reference StereoConvolver surface
        (prepare/loadImpulse/process/clear + bufferLL/RR/LR/RL, isQuad, size)."""
        self.convolver = make_stereo_convolver()
        self.loadConvolver = make_stereo_convolver()
        self.threaded = threaded
        self.loadState = K_IDLE
        self.xfade = 0
        self.xfadelen = 0
        self._worker = None

    def prepare(self, sampleRate: float, samplesPerBlock: int):      # PluginProcessor.cpp:607-614
        self.srate = float(sampleRate)
        self.convolver.prepare(samplesPerBlock)
        self.loadConvolver.prepare(samplesPerBlock)
        self.warmer = np.zeros((2, int(math.ceil(sampleRate)) // 4), np.float32)   # 0.25 s, :610
        self.warmwritepos = 0

    def loadImpulse(self, imp):                                        # :638 (initial, synchronous)
        self.convolver.loadImpulse(imp)

    def request_impulse(self, imp) -> bool:                            # :1675-1691
        if self.loadState != K_IDLE:
            return False
        self.loadState = K_LOADING

        def job():
            self.loadConvolver.loadImpulse(imp)
            self.loadState = K_READY

        if self.threaded:
            self._worker = threading.Thread(target=job)
            self._worker.start()
        else:
            job()
        return True

    def wait_loaded(self):
        if self._worker is not None:
            self._worker.join()
            self._worker = None

    def _warm_write(self, sendL, sendR, n):                            # :1655-1668
        W = self.warmer.shape[1]
        for ch, src in ((0, sendL), (1, sendR)):
            space = W - self.warmwritepos
            if n <= space:
                self.warmer[ch, self.warmwritepos:self.warmwritepos + n] = src[:n]
            else:
                self.warmer[ch, self.warmwritepos:] = src[:space]
                self.warmer[ch, :n - space] = src[space:n]
        self.warmwritepos = (self.warmwritepos + n) % W

    def _warm_up(self):                                                # :1695-1755
        W = self.warmer.shape[1]
        size = self.convolver.size
        numBlocks = W // size
        start = (self.warmwritepos + 1) % W
        idx = (start + np.arange(numBlocks * size)) % W
        replay = self.warmer[:, idx]
        # one multi-block call; the reference issues numBlocks calls of `size` samples
        if hasattr(self.loadConvolver, "warm"):
            self.loadConvolver.warm(replay[0], replay[1], numBlocks * size, True)
        else:
            for i in range(numBlocks):
                self.loadConvolver.process(replay[0, i * size:], replay[1, i * size:], size, True)
        self.loadState = K_FADING
        self.xfade = int(math.ceil(self.srate * CONV_XFADE_MS / 1000.0))
        self.xfadelen = self.xfade

    def process(self, sendL, sendR, delayedL, delayedR, numSamples: int, tsenabled: bool = True):
        """One host block. send* feeds the warm-up ring and the fading-in convolver, delayed* (the
        pre-delayed send) the current one. Returns the wet bus (L, R) before envelope/width."""
        sendL = np.asarray(sendL, np.float32); sendR = np.asarray(sendR, np.float32)
        self._warm_write(sendL, sendR, numSamples)
        if self.loadState == K_READY:
            self._warm_up()
        conv = self.convolver
        conv.process(delayedL, delayedR, numSamples)                   # :1793-1797
        wet = np.zeros((2, numSamples), np.float32)
        if self.loadState == K_FADING:                                 # :1800-1830
            load = self.loadConvolver
            load.process(sendL, sendR, numSamples, True)
            nbuf = len(conv.bufferLL)                                  # the loop runs over the whole buffer
            xf = self.xfade - np.arange(nbuf, dtype=np.float32)
            alpha = np.clip(np.float32(1.0) - xf / np.float32(self.xfadelen), 0.0, 1.0).astype(np.float32)
            conv.bufferLL[:nbuf] *= (np.float32(1.0) - alpha); conv.bufferRR[:nbuf] *= (np.float32(1.0) - alpha)
            load.bufferLL[:nbuf] *= alpha; load.bufferRR[:nbuf] *= alpha
            if conv.isQuad and tsenabled:
                conv.bufferLR[:nbuf] *= (np.float32(1.0) - alpha); conv.bufferRL[:nbuf] *= (np.float32(1.0) - alpha)
            self.xfade -= nbuf
            if self.xfade <= 0:
                self.loadState = K_IDLE
                self.convolver, self.loadConvolver = self.loadConvolver, self.convolver
            wet[0] += self.loadConvolver.bufferLL[:numSamples]
            wet[1] += self.loadConvolver.bufferRR[:numSamples]
        conv = self.convolver                                          # :1833-1838 (after a possible swap)
        wet[0] += conv.bufferLL[:numSamples]
        wet[1] += conv.bufferRR[:numSamples]
        if conv.isQuad and tsenabled:
            wet[0] += conv.bufferRL[:numSamples]
            wet[1] += conv.bufferLR[:numSamples]
        return wet


def wet_bus(wet, yrev, width: float, drygain: float, wetgain: float, dry):
    """Reverb envelope, mid/side width and dry/wet mix (src/PluginProcessor.cpp:1840-1876)."""
    lin = wet[0] * yrev
    rin = wet[1] * yrev
    mid = (lin + rin) * np.float32(0.5)
    side = (lin - rin) * np.float32(0.5)
    norm = np.float32(1.0) / (np.float32(1.0) + np.float32(width))    # float arithmetic, :1842
    lout = (mid + side * np.float32(width)) * norm
    rout = (mid - side * np.float32(width)) * norm
    return np.stack([dry[0] * np.float32(drygain) + lout * np.float32(wetgain),
                     dry[1] * np.float32(drygain) + rout * np.float32(wetgain)]).astype(np.float32)


def wet_mix_device(cur, load=None, xfade: int = 0, xfadelen: int = 1, yrev=None, width: float = 1.0,
                   drygain: float = 1.0, wetgain: float = 1.0, dry=None, out=None, stream=None, device: int = 0):
    """The same epilogue on the DEVICE for blocks that never leave it (rvc_wet_mix_device; kernel
    k_wet_mix): crossfade (per-sample alpha) + true-stereo sum + envelope + width + dry/wet in one pass.
    cur: 2 (LL, RR) or 4 (LL, RR, LR, RL) torch CUDA float32 vectors of the current convolver;
    load: (LL, RR) of the fading-in convolver while a crossfade runs (xfade = countdown at sample 0);
    dry: (L, R), or None to stop after the width stage (the wet bus itself: no gains applied). Returns the two
    output tensors. Asynchronous on `stream` (a raw hipStream_t; default: torch's CURRENT stream on `device`,
    the one _Pair.run has ordered behind the convolvers' streams)."""
    import torch
    from . import _lib as L
    if stream is None:
        stream = torch.cuda.current_stream(device).cuda_stream
    n = cur[0].numel()
    out = out or (torch.empty_like(cur[0]), torch.empty_like(cur[0]))
    p = L.WetParams()
    ptr = lambda t: t.data_ptr() if t is not None else None
    for i in range(4):
        p.cur[i] = ptr(cur[i]) if i < len(cur) else None
    for i in range(2):
        p.load[i] = ptr(load[i]) if load is not None else None
        p.dry[i] = ptr(dry[i]) if dry is not None else None
        p.out[i] = ptr(out[i])
    p.xfade, p.xfadelen = int(xfade), int(xfadelen)
    p.yrev = ptr(yrev)
    p.width, p.drygain, p.wetgain, p.n = width, drygain, wetgain, n
    import ctypes as C
    if not L.lib().rvc_wet_mix_device(device, stream, C.byref(p)):
        raise RuntimeError("rvc_wet_mix_device failed")
    return out


def send_pre_device(inp, ysend=None, delay_ring=None, delaypos: int = 0, predelay: int = 0, warm_ring=None,
                    warmwritepos: int = 0, want_send: bool = True, stream=None, device: int = 0):
    """The send pre-stage on the DEVICE (rvc_send_pre_device; kernel k_send_pre): send envelope multiply
    (src/PluginProcessor.cpp:1640-1653, IIR send filters excluded), warm-up ring write (:1655-1668), pre-delay
    ring write + delayed read (:1766-1790). inp: (2, n) torch CUDA float32; delay_ring / warm_ring: (2, size)
    tensors updated in place. Returns (send, delayed) as (2, n) tensors (None where not requested); the caller
    advances delaypos / warmwritepos by n modulo the ring sizes."""
    import ctypes as C
    import torch
    from . import _lib as L
    if stream is None:
        stream = torch.cuda.current_stream(device).cuda_stream
    n = inp.shape[1]
    assert inp.is_cuda and inp.dtype == torch.float32 and inp.shape[0] == 2 and inp.stride(1) == 1
    send = torch.empty((2, n), device=inp.device) if want_send else None
    delayed = torch.empty((2, n), device=inp.device) if delay_ring is not None else None
    p = L.SendParams()
    for c in range(2):
        p.in_[c] = inp[c].data_ptr()
        p.send[c] = send[c].data_ptr() if send is not None else None
        p.delay_ring[c] = delay_ring[c].data_ptr() if delay_ring is not None else None
        p.delayed[c] = delayed[c].data_ptr() if delayed is not None else None
        p.warm_ring[c] = warm_ring[c].data_ptr() if warm_ring is not None else None
    p.ysend = ysend.data_ptr() if ysend is not None else None
    p.delay_size = delay_ring.shape[1] if delay_ring is not None else 0
    p.delaypos, p.predelay = int(delaypos), int(predelay)
    p.warm_size = warm_ring.shape[1] if warm_ring is not None else 0
    p.warmwritepos = int(warmwritepos)
    p.n = n
    if not L.lib().rvc_send_pre_device(device, stream, C.byref(p)):
        raise RuntimeError("rvc_send_pre_device failed")
    return send, delayed


class DeviceHotSwap:
    """The same convolver section of processBlock (src/PluginProcessor.cpp:1655-1756, 1793-1876) for
    blocks that live on the DEVICE: torch CUDA tensors in, torch CUDA tensors out, no host round trip.
    Two device StereoConvolver pairs (current / loading), the 0.25 s warm-up ring in HBM, warm-up as ONE
    multi-block rvc_set_process_device call per pair, and crossfade + true-stereo sum + envelope + width +
    dry/wet in ONE pass (rvc_wet_mix_device). Impulses are reevr_amd.Impulse objects (device-resident,
    rvc_set_init_impulse) or any object with bufferLL/RR[/LR/RL] + isQuad. Everything is enqueued on the
    current pair's foreground stream and the streams are chained with torch stream dependencies."""

    class _Pair:
        def __init__(self, device):
            from .convolver import ConvolverSet
            self.main = ConvolverSet(2, device)        # LL, RR
            self.cross = ConvolverSet(2, device)       # LR, RL
            self.isQuad = False

        def load(self, imp, head, tail, max_len):
            if getattr(imp, "_h", None) and hasattr(imp, "device_ptr"):
                assert self.main.init_impulse(head, tail, imp, [0, 1], max_len)
                self.isQuad = bool(imp.isQuad)
                if self.isQuad:
                    assert self.cross.init_impulse(head, tail, imp, [2, 3], max_len)
            else:
                assert self.main.init(head, tail, [imp.bufferLL, imp.bufferRR], max_len)
                self.isQuad = bool(imp.isQuad)
                if self.isQuad:
                    assert self.cross.init(head, tail, [imp.bufferLR, imp.bufferRL], max_len)

        def run(self, x, force2=False):
            """x: (2, n) device tensor. Returns [LL, RR] or [LL, RR, LR, RL] (device vectors); synchronous
            with torch's current stream on return (ordering, not a host wait)."""
            y = self.main.process_device(x, sync=False)
            out = [y[0], y[1]]
            if self.isQuad and not force2:
                z = self.cross.process_device(x, sync=False)
                out += [z[0], z[1]]
            return out

    def __init__(self, device: int = 0):
        self.device = device
        self.cur = self._Pair(device)
        self.nxt = self._Pair(device)
        self.loadState = K_IDLE
        self.xfade = self.xfadelen = 0
        self._worker = None

    def prepare(self, sampleRate: float, samplesPerBlock: int):
        import torch
        self.srate = float(sampleRate)
        self.size = int(samplesPerBlock)
        self.head = 1
        while self.head < self.size:
            self.head *= 2
        self.tail = max(8192, 2 * self.head)
        self.W = int(math.ceil(sampleRate)) // 4                     # 0.25 s warm-up ring, :610
        self.warmer = torch.zeros(2, self.W, device=f"cuda:{self.device}")
        self.warmwritepos = 0
        self.delay_ring = torch.zeros(2, int(2.0 * sampleRate), device=f"cuda:{self.device}")   # delayBuffer, :640
        self.delaypos = 0
        self.max_len = max(self.size, (self.W // self.size) * self.size)

    def loadImpulse(self, imp):
        self.cur.load(imp, self.head, self.tail, self.max_len)

    def request_impulse(self, imp, threaded: bool = False) -> bool:
        if self.loadState != K_IDLE:
            return False
        self.loadState = K_LOADING

        def job():
            self.nxt.load(imp, self.head, self.tail, self.max_len)
            self.loadState = K_READY
        if threaded:
            self._worker = threading.Thread(target=job)
            self._worker.start()
        else:
            job()
        return True

    def wait_loaded(self):
        if self._worker is not None:
            self._worker.join()
            self._worker = None

    def process_input(self, inp, ysend=None, predelay: int = 0, **kw):
        """The whole convolver section from the plug-in's INPUT block (2, n), all on the device: send pre-stage
        (rvc_send_pre_device: envelope, warm-up ring, pre-delay ring; src/PluginProcessor.cpp:1640-1668,
        1766-1790) and then process(). The dry signal of the mix defaults to the input itself."""
        n = inp.shape[1]
        send, delayed = send_pre_device(inp, ysend, self.delay_ring, self.delaypos, predelay, self.warmer,
                                        self.warmwritepos, device=self.device)
        self.delaypos = (self.delaypos + n) % self.delay_ring.shape[1]
        self.warmwritepos = (self.warmwritepos + n) % self.W
        kw.setdefault("dry", (inp[0], inp[1]))
        return self.process(send, delayed, _warm_written=True, **kw)

    def process(self, send, delayed, yrev=None, width: float = 1.0, drygain: float = 0.0, wetgain: float = 1.0,
                dry=None, tsenabled: bool = True, _warm_written: bool = False):
        """send / delayed: (2, n) device tensors (send feeds the warm-up ring and the fading-in
        convolver, delayed the current one). Returns (outL, outR) device tensors."""
        import torch
        n = send.shape[1]
        if not _warm_written:                                        # warm-up ring write (:1655-1668)
            idx = (self.warmwritepos + torch.arange(n, device=send.device)) % self.W
            self.warmer[:, idx] = send
            self.warmwritepos = (self.warmwritepos + n) % self.W
        if self.loadState == K_READY:                                # warm-up replay, :1695-1755
            numBlocks = self.W // self.size
            start = (self.warmwritepos + 1) % self.W
            ridx = (start + torch.arange(numBlocks * self.size, device=send.device)) % self.W
            self.nxt.run(self.warmer[:, ridx].contiguous(), force2=True)     # one multi-block call, output discarded
            self.loadState = K_FADING
            self.xfade = int(math.ceil(self.srate * CONV_XFADE_MS / 1000.0))
            self.xfadelen = self.xfade
        cur = self.cur.run(delayed.contiguous())                     # :1793-1797
        if not (self.cur.isQuad and tsenabled):
            cur = cur[:2]
        load = None
        xf = self.xfade
        if self.loadState == K_FADING:                               # :1800-1830
            load = self.nxt.run(send.contiguous(), force2=True)[:2]
            self.xfade -= self.size                                   # the reference's loop runs over the whole buffer
        out = wet_mix_device(cur, load=load, xfade=xf, xfadelen=max(self.xfadelen, 1), yrev=yrev, width=width,
                             drygain=drygain, wetgain=wetgain, dry=dry, device=self.device)
        if self.loadState == K_FADING and self.xfade <= 0:
            self.loadState = K_IDLE
            self.cur, self.nxt = self.nxt, self.cur
        return out
