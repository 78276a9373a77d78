"""Oracle-side restatements of the two elementwise stages either side of the convolvers in
REEVRAudioProcessor::processBlock, statement for statement like the reference's scalar loops
(float32 at every step). TEST INFRASTRUCTURE ONLY -- the product's own forms are
reevr_amd.hotswap.wet_bus (host) and the kernels k_wet_mix / k_send_pre (device).

  ref_wet_bus    src/PluginProcessor.cpp:1840-1876   reverb envelope, mid/side width, dry/wet mix
  ref_send_pre   src/PluginProcessor.cpp:1640-1668, 1766-1790   send envelope, warm-up ring, pre-delay ring
"""
import numpy as np

f32 = np.float32


def ref_wet_bus(wet, yrev, width, drygain, wetgain, dry):
    """wet, dry: (2, n) float32; yrev: (n,). The non-monitor branch (:1860-1876)."""
    wet = np.asarray(wet, f32)
    n = wet.shape[1]
    out = np.zeros((2, n), f32)
    normalization = f32(1.0) / (f32(1.0) + f32(width))                  # :1842
    for s in range(n):                                                  # :1843-1857
        lin = f32(wet[0, s] * f32(yrev[s]))
        rin = f32(wet[1, s] * f32(yrev[s]))
        mid = f32(f32(lin + rin) * f32(0.5))
        side = f32(f32(lin - rin) * f32(0.5))
        lout = f32(f32(mid + f32(side * f32(width))) * normalization)
        rout = f32(f32(mid - f32(side * f32(width))) * normalization)
        if dry is None:
            out[0, s], out[1, s] = lout, rout
        else:                                                           # applyGain both, then addFrom (:1861-1875)
            out[0, s] = f32(f32(dry[0][s] * f32(drygain)) + f32(lout * f32(wetgain)))
            out[1, s] = f32(f32(dry[1][s] * f32(drygain)) + f32(rout * f32(wetgain)))
    return out


class RefSendPre:
    """The send envelope, warmer and pre-delay rings of processBlock with the reference's own loops
    (IIR send filters off: irLowcut <= 20, irHighcut >= 20000)."""

    def __init__(self, delay_size, warm_size):
        self.delayBuffer = np.zeros((2, delay_size), f32)
        self.warmer = np.zeros((2, warm_size), f32)
        self.delaypos = 0
        self.warmwritepos = 0

    def process(self, L, R, ysend, predelay):
        n = len(L)
        send = np.zeros((2, n), f32)
        for s in range(n):                                              # :1640-1653
            send[0, s] = f32(f32(L[s]) * f32(ysend[s]))
            send[1, s] = f32(f32(R[s]) * f32(ysend[s]))
        W = self.warmer.shape[1]                                        # :1655-1668 (copyFrom in <= 2 pieces)
        space = W - self.warmwritepos
        for ch in range(2):
            if n <= space:
                self.warmer[ch, self.warmwritepos:self.warmwritepos + n] = send[ch]
            else:
                self.warmer[ch, self.warmwritepos:] = send[ch, :space]
                self.warmer[ch, :n - space] = send[ch, space:]
        self.warmwritepos = (self.warmwritepos + n) % W
        size = self.delayBuffer.shape[1]                                # :1766-1790
        for ch in range(2):
            for i in range(n):
                self.delayBuffer[ch, (self.delaypos + i) % size] = send[ch, i]
        delayed = np.zeros((2, n), f32)
        for ch in range(2):
            readPosition = (self.delaypos + size - predelay) % size
            for i in range(n):
                delayed[ch, i] = self.delayBuffer[ch, (readPosition + i) % size]
        self.delaypos = (self.delaypos + n) % size
        return send, delayed
