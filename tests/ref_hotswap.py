"""Oracle-side restatement of the reference's convolver hot-swap (src/PluginProcessor.cpp:
1655-1756, 1793-1838) on top of oracle convolvers, call for call like the reference: the warm-up
is a loop of block-sized process() calls. TEST INFRASTRUCTURE ONLY."""
import math

import numpy as np

from oracle import oracle_py as O


class OracleStereoConvolver:
    """src/dsp/StereoConvolver.cpp on four oracle TwoStageFFTConvolvers."""

    def __init__(self, which="orc"):
        self.c = {k: O.TwoStageFFTConvolver(which) for k in ("LL", "RR", "LR", "RL")}
        self.size = 0
        self.isQuad = False

    def prepare(self, n):
        self.size = n
        self.head = 1
        while self.head < n:
            self.head *= 2
        self.tail = max(8192, 2 * self.head)
        for k in self.c:
            setattr(self, "buffer" + k, np.zeros(n, np.float32))

    def loadImpulse(self, imp):
        self.c["LL"].init(self.head, self.tail, imp.bufferLL)
        self.c["RR"].init(self.head, self.tail, imp.bufferRR)
        self.isQuad = bool(imp.isQuad)
        if self.isQuad:
            self.c["LR"].init(self.head, self.tail, imp.bufferLR)
            self.c["RL"].init(self.head, self.tail, imp.bufferRL)

    def process(self, L, R, n, force2Chans=False):
        self.bufferLL[:n] = self.c["LL"].process(L[:n])
        self.bufferRR[:n] = self.c["RR"].process(R[:n])
        if self.isQuad and not force2Chans:
            self.bufferLR[:n] = self.c["LR"].process(L[:n])
            self.bufferRL[:n] = self.c["RL"].process(R[:n])


class RefHotSwap:
    def __init__(self):
        self.convolver = OracleStereoConvolver()
        self.loadConvolver = OracleStereoConvolver()
        self.state = 0
        self.xfade = self.xfadelen = 0

    def prepare(self, sr, n):
        self.srate = float(sr)
        self.convolver.prepare(n); self.loadConvolver.prepare(n)
        self.warmer = np.zeros((2, int(math.ceil(sr)) // 4), np.float32)
        self.warmwritepos = 0

    def loadImpulse(self, imp):
        self.convolver.loadImpulse(imp)

    def request_impulse(self, imp):
        self.loadConvolver.loadImpulse(imp)
        self.state = 2   # kReady

    def process(self, sendL, sendR, delL, delR, n, ts=True):
        W = self.warmer.shape[1]
        for i in range(n):                                   # :1655-1668
            self.warmer[0, (self.warmwritepos + i) % W] = sendL[i]
            self.warmer[1, (self.warmwritepos + i) % W] = sendR[i]
        self.warmwritepos = (self.warmwritepos + n) % W
        if self.state == 2:                                  # :1695-1755
            size = self.convolver.size
            start = (self.warmwritepos + 1) % W
            for _ in range(W // size):
                idx = (start + np.arange(size)) % W
                self.loadConvolver.process(self.warmer[0, idx], self.warmer[1, idx], size, True)
                start = (start + size) % W
            self.state = 3
            self.xfade = int(math.ceil(self.srate * 50 / 1000.0))
            self.xfadelen = self.xfade
        self.convolver.process(delL, delR, n)
        wet = np.zeros((2, n), np.float32)
        if self.state == 3:
            self.loadConvolver.process(sendL, sendR, n, True)
            cv, ld = self.convolver, self.loadConvolver
            for i in range(len(cv.bufferLL)):
                alpha = np.float32(min(max(1.0 - np.float32(self.xfade) / np.float32(self.xfadelen), 0.0), 1.0))
                cv.bufferLL[i] *= np.float32(1) - alpha; cv.bufferRR[i] *= np.float32(1) - alpha
                ld.bufferLL[i] *= alpha; ld.bufferRR[i] *= alpha
                if cv.isQuad and ts:
                    cv.bufferLR[i] *= np.float32(1) - alpha; cv.bufferRL[i] *= np.float32(1) - alpha
                self.xfade -= 1
            if self.xfade <= 0:
                self.state = 0
                self.convolver, self.loadConvolver = self.loadConvolver, self.convolver
            wet[0] += self.loadConvolver.bufferLL[:n]; wet[1] += self.loadConvolver.bufferRR[:n]
        cv = self.convolver
        wet[0] += cv.bufferLL[:n]; wet[1] += cv.bufferRR[:n]
        if cv.isQuad and ts:
            wet[0] += cv.bufferRL[:n]; wet[1] += cv.bufferLR[:n]
        return wet
