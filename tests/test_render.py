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


def _rel_rms(got, want):
    got, want = got.astype(np.float64), want.astype(np.float64)
    return np.sqrt(np.mean((got - want) ** 2)) / max(np.sqrt(np.mean(want ** 2)), 1e-30)


@pytest.mark.gpu
def test_raw_batch_render_config5_against_pinned_oracle():
    """f-4 as SURVEY 8f wrote it: the many-channel batch renderer in raw mode on BASELINE config 5's workload -- 64 parallel
    mono channels, 5 s IR @ 48 kHz each, block 4096 (head 4096 / tail 8192) -- is N x TwoStageFFTConvolver::process
    (TwoStageFFTConvolver.cpp:151-233): every channel against the PINNED oracle, 1e-5 RMS."""
    from oracle import oracle_py as O
    from reevr_amd.render import render_raw
    nch, ir_len, frames = 64, 240000, 10 * 48000
    irs = [synth.synth_ir(ir_len, 2, inst=u)[c] for u in range(nch // 2) for c in range(2)]
    x = np.stack([synth.synth_input(frames, c) for c in range(nch)])
    y = render_raw(x, irs, block=4096)
    assert y.shape == (nch, frames) and np.isfinite(y).all()
    for c in range(nch):
        o = O.TwoStageFFTConvolver()
        assert o.init(4096, 8192, irs[c])
        assert _rel_rms(y[c], o.process(x[c])) <= 1e-5, c


@pytest.mark.gpu
def test_raw_batch_render_cli_ragged(tmp_path):
    """The CLI in raw mode on a ragged small case: two input files of different lengths and channel counts (3 channels in
    all), a 3-channel impulse file whose channels end at different points (one is silent: an empty impulse gives silence,
    TwoStageFFTConvolver.cpp:112-115), a block that is not a power of two, the reverb ringing out (--tail), chunked calls."""
    from oracle import oracle_py as O
    from reevr_amd import render
    sr = 44100
    a = np.stack([synth.synth_input(7001, 40), synth.synth_input(7001, 41)])
    b = synth.synth_input(5000, 42)[None, :]
    m = 3001
    t = np.arange(m) / m
    ir = np.zeros((3, m), np.float32)
    ir[0] = synth.white_noise(m, 7) * np.exp(-5.0 * t)
    ir[1, :777] = synth.white_noise(777, 8) * 0.3
    pa, pb, pir, pout = (str(tmp_path / f) for f in ("a.wav", "b.wav", "ir.wav", "out.wav"))
    write_wav(pa, a, sr)
    write_wav(pb, b, sr)
    write_wav(pir, ir, sr)
    assert render.main(["--raw", "--ir", pir, "--in", pa, pb, "--out", pout, "--block", "100", "--tail"]) == 0
    y, osr = read_wav(pout)
    assert osr == sr and y.shape == (3, 7001 + m)
    x = np.zeros((3, 7001 + m), np.float32)
    x[:2, :7001] = a
    x[2, :5000] = b[0]
    for c in range(3):
        o = O.TwoStageFFTConvolver()
        assert o.init(100, 8192, ir[c])
        want = o.process(x[c])
        if c == 2:
            assert np.all(want == 0) and np.all(y[c] == 0)
        else:
            assert _rel_rms(y[c], want) <= 1e-5, c
    # ONE impulse channel shared by every input channel; two calls of the chunked path
    from reevr_amd.render import render_raw
    y1 = render_raw(x[:, :7001], [ir[0]], block=512, chunk=4000)
    for c in range(3):
        o = O.TwoStageFFTConvolver()
        assert o.init(512, 8192, ir[0])
        assert _rel_rms(y1[c], o.process(x[c, :7001])) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_raw_batch_render_sharded_over_ranks(tmp_path, world):
    """`torch.distributed.run -m reevr_amd.render --raw`: the channels dealt to the ranks (shard.units_for_rank, unequal
    shards padded with silent channels), one gather per chunk, rank 0 writes the file -- against the pinned oracle. With
    fewer devices than ranks the ranks share device 0 over gloo (REEVR_BENCH_SAME_DEVICE=1), else one device each over RCCL."""
    import socket
    import subprocess
    from oracle import oracle_py as O
    from reevr_amd import _lib
    sr, nch, frames, m = 48000, 5, 30000, 20000
    x = np.stack([synth.synth_input(frames, 60 + c) for c in range(nch)])
    t = np.arange(m) / m
    ir = np.stack([(0.4 * synth.white_noise(m, 500 + c) * np.exp(-6.0 * t)).astype(np.float32) for c in range(nch)])
    pin, pir, pout = (str(tmp_path / f) for f in ("in.wav", "ir.wav", "out.wav"))
    write_wav(pin, x, sr)
    write_wav(pir, ir, sr)
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    if _lib.lib().rvc_device_count() < world:
        env["REEVR_BENCH_SAME_DEVICE"] = "1"
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "reevr_amd.render", "--raw", "--ir", pir, "--in", pin, "--out", pout, "--block", "512"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    y, osr = read_wav(pout)
    assert osr == sr and y.shape == (nch, frames)
    for c in range(nch):
        o = O.TwoStageFFTConvolver()
        assert o.init(512, 8192, ir[c])
        assert _rel_rms(y[c], o.process(x[c])) <= 1e-5, c
