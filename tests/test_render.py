"""Offline renderer: WAV round trips on the CPU, and on the GPU the whole
file -> impulse stages -> convolvers -> wet bus -> file chain against the CPU oracles."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from reevr_amd import synth  # noqa: E402
from reevr_amd.wavio import read_wav, write_wav  # noqa: E402


def test_wav_round_trips(tmp_path):
    x = np.stack([synth.white_noise(1000, 5), 0.25 * synth.white_noise(1000, 6), synth.ramp(1000) * 1e-3])
    p = str(tmp_path / "f.wav")
    write_wav(p, x, 44100)
    y, sr = read_wav(p)
    assert sr == 44100 and np.array_equal(x, y)                 # float32: exact
    write_wav(p, x[:2], 48000, float32=False)
    y, sr = read_wav(p)
    assert sr == 48000 and y.shape == (2, 1000) and np.max(np.abs(y - x[:2])) <= 2.0 / 32768   # quantisation + the 32767/32768 convention
    # 24-bit PCM written by hand
    import struct
    v = np.array([0, 1, -1, 8388607, -8388608, 123456], np.int32)
    payload = b"".join(struct.pack("<i", int(s))[:3] for s in v)
    fmt = struct.pack("<HHIIHH", 1, 1, 8000, 24000, 3, 24)
    with open(p, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(payload)) + b"WAVE" + b"fmt " +
                struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload)
    y, sr = read_wav(p)
    assert sr == 8000 and np.allclose(y[0], v / 8388608.0, atol=1e-7)
    with pytest.raises(ValueError):
        open(p, "wb").write(b"not a wav file at all")
        read_wav(p)


@pytest.mark.gpu
@pytest.mark.parametrize("nir", [1, 2, 4])
def test_render_matches_oracle_chain(tmp_path, nir):
    from oracle import oracle_py as O
    from reevr_amd import render
    sr, n, m = 48000, 30000, 20000
    x = np.stack([synth.synth_input(n, c) for c in range(2)])
    t = np.arange(m) / m
    ir = np.stack([(0.5 * synth.white_noise(m, 900 + c) * np.exp(-6.9 * t)).astype(np.float32) for c in range(nir)])
    pin, pir, pout = (str(tmp_path / f) for f in ("in.wav", "ir.wav", "out.wav"))
    write_wav(pin, x, sr)
    write_wav(pir, ir, sr)
    assert render.main(["--ir", pir, "--in", pin, "--out", pout, "--wet", "0.7", "--dry", "0.2", "--tail",
                        "--attack", "0.02", "--decay", "0.5", "--gain", "1.5"]) == 0
    y, osr = read_wav(pout)
    assert osr == sr and y.shape[0] == 2
    # CPU chain: impulse restatement -> TwoStage oracle -> wet bus
    raw = [ir[0], ir[3], ir[1], ir[2]] if nir == 4 else ([ir[0], ir[1]] if nir == 2 else [ir[0], ir[0]])
    from reevr_amd import Impulse
    end = Impulse.tail_start(raw)                       # Impulse::load's trailing-silence trim (|x| < 1e-3)
    assert 0 < end < m
    raw = [r[:end] for r in raw]
    imp = O.impulse_recalc(raw, attack=0.02, decay=0.5, gain=1.5, srate=float(sr))["buffers"]
    assert y.shape[1] == n + imp[0].size
    xin = np.concatenate([x, np.zeros((2, imp[0].size), np.float32)], axis=1)

    def conv(h, sig):
        c = O.TwoStageFFTConvolver()
        assert c.init(512, 8192, h)
        return c.process(sig)
    wl = conv(imp[0], xin[0]).astype(np.float64)            # LL <- L
    wr = conv(imp[1], xin[1]).astype(np.float64)            # RR <- R
    if nir == 4:
        wl += conv(imp[3], xin[1])                          # RL <- R goes to the left bus
        wr += conv(imp[2], xin[0])                          # LR <- L goes to the right bus
    from tests.ref_wetbus import ref_wet_bus as wet_bus   # oracle-side restatement of PluginProcessor.cpp:1840-1876
    want = wet_bus(np.stack([wl, wr]).astype(np.float32), np.ones(xin.shape[1], np.float32), 1.0, 0.2, 0.7, xin)
    err = np.sqrt(np.mean((y - want) ** 2))
    assert err <= 1e-5, err
