"""GPU parity tests (-m gpu): the HIP engine, called through the C ABI, against
  * the committed golden fixtures generated from the untouched reference (tests/golden),
  * the reference's 58 known-answer cases with its own pass rule (Test.cpp:129-145),
  * the CPU oracle (oracle/) on the same seeded inputs, full length,
  * size-independent properties at BASELINE.json's full sizes.
Tolerance: RMS(out - ref) <= 1e-5 relative to the output RMS (north_star: 1e-5 RMS, float32).
"""
import os

import numpy as np
import pytest

import reevr_amd
from oracle import oracle_py as O
from reevr_amd import synth
from tests import cases
from tests.conftest import fixture_of

pytestmark = pytest.mark.gpu
TOL = 1e-5


def gpu_factory(kind):
    return reevr_amd.FFTConvolver() if kind == "fftconv" else reevr_amd.TwoStageFFTConvolver()


def gpu64_factory(kind):
    """RVC_FLAG_FFT_F64: transforms in double, spectra in float -- the reference's precision."""
    return (reevr_amd.FFTConvolver(fft_f64=True) if kind == "fftconv"
            else reevr_amd.TwoStageFFTConvolver(fft_f64=True))


def gpu_bg_factory(kind):
    return reevr_amd.FFTConvolver() if kind == "fftconv" else reevr_amd.Convolver()


def orc_factory(kind):
    return O.FFTConvolver("orc") if kind == "fftconv" else O.TwoStageFFTConvolver("orc")


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def test_gpu_present():
    from reevr_amd import _lib
    assert _lib.lib().rvc_device_count() >= 1


KATS = [("fftconv", t) for t in cases.KAT_FFTCONV] + [("twostage", t) for t in cases.KAT_TWOSTAGE]


def gpu32_factory(kind):
    """RVC_FLAG_FFT_F32: float transforms whatever the set size (the default of large lock-step sets)."""
    return (reevr_amd.FFTConvolver(fft_f32=True) if kind == "fftconv"
            else reevr_amd.TwoStageFFTConvolver(fft_f32=True))


class _LaneOfManyChannels:
    """The reference's per-object surface on channel 0 of a 12-channel set -- more than 8 channels: the precision policy of the
    lock-step sets (stages with partitions of 2048 .. 8192 samples run their INVERSE transform in double, rvc.h RVC_FLAG_FFT_F64);
    LONG: created with RVC_FLAG_FFT_F64_LONG (both transforms of those stages in double, the small sets' rule)."""
    NCH = 12
    LONG = False

    def __init__(self, kind):
        self.kind = kind
        self._set = reevr_amd.ConvolverSet(self.NCH, fft_f64_long=self.LONG)

    def init(self, *a):
        ir = np.asarray(a[-1], np.float32)
        irs = [ir * np.float32(1.0 - 0.05 * c) for c in range(self.NCH)]
        return self._set.init_uniform(a[0], irs) if self.kind == "fftconv" else self._set.init(a[0], a[1], irs)

    def process(self, x):
        x = np.asarray(x, np.float32).reshape(-1)
        return self._set.process(np.stack([x * np.float32(1.0 - 0.03 * c) for c in range(self.NCH)]))[0]


class _LaneOfManyChannelsLong(_LaneOfManyChannels):
    LONG = True


@pytest.mark.parametrize("mode", ["default", "f64", "f32", "default_12ch", "f64_long_12ch"])
@pytest.mark.parametrize("kind,tup", KATS, ids=[cases.kat_name(k, t) for k, t in KATS])
def test_kat_vs_golden_and_reference_rule(golden, kind, tup, mode):
    """The reference's 58 known-answer cases (Test.cpp:256-329) in the three precision modes.
    Parity bar (north_star): RMS error <= 1e-5 of the output RMS -- every mode, all cases.
    The reference's own pass rule (Test.cpp:129-145, margin < 1): all 58 cases in the DEFAULT mode (sets of up to 8
    channels run stages with partitions of 2048 .. 8192 samples in double, rvc.h RVC_FLAG_FFT_F64) and with every
    transform in double. With float transforms throughout (RVC_FLAG_FFT_F32, what large lock-step sets run) the rule
    holds for partitions below 2048; with 2048-sample partitions of a 0.1*(i+1) ramp the first ~100 outputs (values
    1..1700) share a 4096-point transform with values of 1.5e7, and a float32 FFT's 2e-7 relative noise is then ~1.3
    absolute against the rule's 1.234: margin 1.01 .. 1.10 measured on MI355X -- bounded here at 1.15.
    default_12ch: a set of MORE than 8 channels with no precision flag -- what lock-step sets run by default since round 5: the
    INVERSE transform of stages with partitions of 2048 .. 8192 samples in double (the float noise that breaks the rule is the
    inverse transform's; measured margins 0.03 .. 0.16 on the four cases float fails) -- meets the rule on all 58 cases (limit 1.0);
    f64_long_12ch: the same set with RVC_FLAG_FFT_F64_LONG (both transforms of those stages in double) too."""
    factory = {"default": gpu_factory, "f64": gpu64_factory, "f32": gpu32_factory, "default_12ch": _LaneOfManyChannels,
               "f64_long_12ch": _LaneOfManyChannelsLong}[mode]
    out = cases.run_kat(factory, kind, tup)
    cases.compare_to_fixture(out, fixture_of(golden["kat"], cases.kat_name(kind, tup)), TOL)
    exact = O.direct_convolve(synth.ramp(tup[0]), synth.ramp(tup[1]))
    margin = cases.kat_margin(out, exact, tup[1])
    block = tup[4]          # (two-stage cases: IR < 2 x tail block, every partition is head-sized)
    limit = 1.15 if (mode == "f32" and block >= 2048) else 1.0
    assert margin < limit, f"margin {margin:.3f}"
    if limit == 1.0:
        assert cases.kat_tolerance_ok(out, exact, tup[1])


@pytest.mark.parametrize("name", list(cases.SYNTH_CASES))
def test_synth_vs_golden(golden, name):
    out = cases.run_synth_case(gpu_factory, cases.SYNTH_CASES[name])
    for c in range(out.shape[0]):
        cases.compare_to_fixture(out[c], fixture_of(golden["synth"], f"{name}/ch{c}"), TOL)


@pytest.mark.parametrize("name", ["small_three_stage", "ragged_calls_b512", "clear_block_aligned",
                                  "cfg2_stereo_10s_b512", "cfg5_5s_b4096", "head_eq_tail", "nonpow2_sizes"])
def test_synth_vs_oracle_full(name):
    case = cases.SYNTH_CASES[name]
    got = cases.run_synth_case(gpu_factory, case)
    want = cases.run_synth_case(orc_factory, case)
    for c in range(got.shape[0]):
        assert rel_rms(got[c], want[c]) <= TOL


@pytest.mark.parametrize("name", ["small_three_stage", "ragged_calls_b512", "cfg4_inst3_10s_b512"])
def test_background_stream_matches(golden, name):
    """Convolver (tail on the second stream, event hand-off) == inline tail."""
    out = cases.run_synth_case(gpu_bg_factory, cases.SYNTH_CASES[name])
    for c in range(out.shape[0]):
        cases.compare_to_fixture(out[c], fixture_of(golden["synth"], f"{name}/ch{c}"), TOL)


@pytest.mark.parametrize("name", ["cfg1_mono_1s_b512", "cfg2_stereo_10s_b512", "cfg3_stereo_30s96k_b256",
                                  "cfg3_full_length", "cfg5_5s_b4096", "small_three_stage"])
def test_one_big_call_equals_block_calls(golden, name):
    """Multi-block fast path: the whole input in ONE process() call (time-tiled FIR, batched
    FFTs) must give the reference's block-by-block result (call-pattern independence)."""
    case = cases.SYNTH_CASES[name]
    irs = cases.make_ir(case["ir"])
    x = cases.make_input(case, irs.shape[0])
    s = reevr_amd.ConvolverSet(irs.shape[0])
    if case["kind"] == "fftconv":
        assert s.init_uniform(case["block"], list(irs), max_len=case["frames"])
    else:
        assert s.init(case["head"], case["tail"], list(irs), max_len=case["frames"])
    out = s.process(x)
    assert s.last_error == 0, s.last_error_string
    for c in range(out.shape[0]):
        cases.compare_to_fixture(out[c], fixture_of(golden["synth"], f"{name}/ch{c}"), TOL)
    # and again split at an awkward point, through the device-pointer entry
    import torch
    s.clear()
    cut = case["frames"] // 3 + 7
    dx = torch.from_numpy(x).cuda()
    o1 = s.process_device(dx[:, :cut].contiguous())
    o2 = s.process_device(dx[:, cut:].contiguous())
    out2 = torch.cat([o1, o2], dim=1).cpu().numpy()
    for c in range(out.shape[0]):
        cases.compare_to_fixture(out2[c], fixture_of(golden["synth"], f"{name}/ch{c}"), TOL)


def test_full_size_properties_cfg2():
    """BASELINE configs[1] at full size (stereo, 10 s IR @ 48 kHz, head 512 / tail 8192, 40 s of
    input in one call): impulse -> IR, linearity, and equality with 512-sample streaming."""
    irs = synth.synth_ir(480000, 2, 0)
    frames = 4 * 480000
    s = reevr_amd.ConvolverSet(2)
    assert s.init(512, 8192, list(irs), max_len=frames)
    assert s.partitions(0) == 32 and s.partitions(1) == 57   # 2*8192/512, ceil((480000-16384)/8192)
    # impulse response identity: delta in -> IR out (exactly the linear-convolution definition)
    d = np.zeros((2, frames), np.float32)
    d[:, 5] = 1.0
    y = s.process(d)
    for c in range(2):
        want = np.zeros(frames, np.float32)
        want[5:5 + 480000] = irs[c]
        assert np.sqrt(np.mean((y[c].astype(np.float64) - want) ** 2)) <= 1e-7
    # linearity: conv(a + 2b) == conv(a) + 2 conv(b)
    a = np.stack([synth.synth_input(frames, c) for c in range(2)])
    b = np.stack([synth.synth_input(frames, 7 + c) for c in range(2)])
    s.clear(); ya = s.process(a)
    s.clear(); yb = s.process(b)
    s.clear(); yab = s.process(a + 2 * b)
    for c in range(2):
        assert rel_rms(yab[c], ya[c].astype(np.float64) + 2.0 * yb[c]) <= TOL
    # streaming in 512-sample calls over the first 2.5 s == the big call
    n = 512 * 240
    t = reevr_amd.ConvolverSet(2, bg_stream=True)
    assert t.init(512, 8192, list(irs))
    ys = np.concatenate([t.process(a[:, i:i + 512]) for i in range(0, n, 512)], axis=1)
    for c in range(2):
        assert rel_rms(ys[c], ya[c, :n]) <= TOL
    # and the oracle on the same first 2.5 s
    for c in range(2):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(512, 8192, irs[c])
        yo = np.concatenate([o.process(a[c, i:i + 512]) for i in range(0, n, 512)])
        assert rel_rms(ya[c, :n], yo) <= TOL


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4", "cfg5"])
def test_full_size_properties_other_configs(cfg):
    """BASELINE configs[2], [3], [4] at full size -- config 3: stereo, 30 s IR @ 96 kHz, block 256 (350 tail
    partitions); config 4: all 8 stereo instances = 16 channels in one set, 10 s IR, block 512; config 5: all
    64 channels, 5 s IR, block 4096 -- through size-independent properties: impulse -> IR, linearity,
    block-synchronous streaming == one big call, and the oracle on a prefix of two channels."""
    import torch
    nch, ir_len, block, n_inst = {"cfg3": (2, 2880000, 256, 1), "cfg4": (16, 480000, 512, 8),
                                  "cfg5": (64, 240000, 4096, 32)}[cfg]
    irs = np.concatenate([synth.synth_ir(ir_len, 2, inst=i) for i in range(n_inst)])
    head = block
    tail = max(8192, 2 * head)
    frames = 2 * ir_len
    frames -= frames % tail
    s = reevr_amd.ConvolverSet(nch)
    assert s.init(block, tail, list(irs), max_len=frames), s.last_error_string
    assert s.partitions(0) == 2 * tail // head
    assert s.partitions(1) == -(-(ir_len - 2 * tail) // tail)
    dev = torch.device("cuda")
    # impulse response identity (delta at sample 5 of every channel)
    d = torch.zeros((nch, frames), device=dev)
    d[:, 5] = 1.0
    y = s.process_device(d).cpu().numpy()
    for c in range(nch):
        want = np.zeros(frames, np.float32)
        want[5:5 + ir_len] = irs[c][:frames - 5]
        assert np.sqrt(np.mean((y[c].astype(np.float64) - want) ** 2)) <= 1e-7, c
    # linearity
    a = torch.from_numpy(np.stack([synth.synth_input(frames, c) for c in range(nch)])).to(dev)
    b = torch.from_numpy(np.stack([synth.synth_input(frames, 100 + c) for c in range(nch)])).to(dev)
    s.clear(); ya = s.process_device(a).cpu().numpy()
    s.clear(); yb = s.process_device(b).cpu().numpy()
    s.clear(); yab = s.process_device(a + 2 * b).cpu().numpy()
    for c in range(nch):
        assert rel_rms(yab[c], ya[c].astype(np.float64) + 2.0 * yb[c]) <= TOL, c
    # block-synchronous streaming (the host's per-block loop, all channels in lock-step) over the whole IR
    # length and a few more tail blocks == the big call
    n = min(frames, (ir_len // tail + 6) * tail)
    for bg in (False, True):
        t = reevr_amd.ConvolverSet(nch, bg_stream=bg)
        assert t.init(block, tail, list(irs), max_len=block)
        ys = t.process_device_blocks(a[:, :n].contiguous(), block).cpu().numpy()
        for c in range(nch):
            assert rel_rms(ys[c], ya[c, :n]) <= TOL, (bg, c)
        t.close()
    # and the oracle, block by block, on a prefix of the first and the last channel
    m = min(n, 40 * tail)
    for c in (0, nch - 1):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(block, tail, irs[c])
        xin = a[c, :m].cpu().numpy()
        yo = np.concatenate([o.process(xin[i:i + block]) for i in range(0, m, block)])
        assert rel_rms(ya[c, :m], yo) <= TOL, c


AUDIOFFT_SIZES = (2, 4, 8, 16, 64, 1024, 16384)


@pytest.mark.parametrize("n,f64", [(n, f) for n in AUDIOFFT_SIZES for f in (0, 1)] + [(32768, 0)])    # (32768: float only, 16384-bin rows)
def test_bare_transforms_vs_audiofft_golden(golden, n, f64):
    """The HIP forward / inverse transform kernels ALONE (rvc_debug_rfft / rvc_debug_irfft: one launch of the
    stage kernels with the stage twiddle tables) against the reference's AudioFFT known answers
    (tests/golden/audiofft.npz, generated from oracle/_ref: AudioFFT.cpp:988-1016 facade, Ooura back-end in
    double). Tolerance, relative to the RMS of the expected vector: 2e-6 with float32 transforms (measured
    ~3e-7), 3e-7 with RVC_FLAG_FFT_F64 (what remains is the float32 rounding of the stored bins)."""
    import ctypes as C
    from reevr_amd import _lib
    L = _lib.lib()
    g = golden["audiofft"]
    x = synth.white_noise(n, 0xF00D + n)
    re = np.empty(n // 2 + 1, np.float32)
    im = np.empty(n // 2 + 1, np.float32)
    fp = lambda a: a.ctypes.data_as(_lib.F32P)
    assert L.rvc_debug_rfft(0, n, f64, fp(x), fp(re), fp(im)) == 1
    tol = 3e-7 if f64 else 2e-6
    want = np.concatenate([g[f"n{n}/re"], g[f"n{n}/im"]]).astype(np.float64)
    got = np.concatenate([re, im]).astype(np.float64)
    scale = np.sqrt(np.mean(want ** 2))
    assert np.sqrt(np.mean((got - want) ** 2)) / scale <= tol
    assert im[0] == 0.0 and im[n // 2] == 0.0                     # AudioFFT.cpp:134-135
    # inverse of the REFERENCE's spectrum == the reference's round trip
    rt = np.empty(n, np.float32)
    wre = np.ascontiguousarray(g[f"n{n}/re"], np.float32)
    wim = np.ascontiguousarray(g[f"n{n}/im"], np.float32)
    assert L.rvc_debug_irfft(0, n, f64, fp(rt), fp(wre), fp(wim)) == 1
    wrt = g[f"n{n}/rt"].astype(np.float64)
    assert np.sqrt(np.mean((rt.astype(np.float64) - wrt) ** 2)) / np.sqrt(np.mean(wrt ** 2)) <= tol
    # argument checks: not a power of two / too large -> 0, nothing written
    assert L.rvc_debug_rfft(0, 24, 0, fp(x), fp(re), fp(im)) == 0 if n >= 24 else True
    assert L.rvc_debug_rfft(0, 1 << 16, 0, fp(x), fp(re), fp(im)) == 0 if n >= (1 << 16) else True


def test_max_block_and_limits():
    ir = synth.synth_ir(40000, 1, 2)[0]
    x = synth.synth_input(60000, 0)
    c = reevr_amd.FFTConvolver()
    assert c.init(16384, ir, max_len=60000)             # 32768-point real FFT in 128 KiB of LDS
    y = c.process(x)
    o = O.FFTConvolver("orc"); assert o.init(16384, ir)
    assert rel_rms(y, o.process(x)) <= TOL
    t = reevr_amd.TwoStageFFTConvolver()
    assert t.init(8192, 16384, ir)                       # host block 8192 -> tail 16384 (StereoConvolver.cpp:11-15)
    y = np.concatenate([t.process(x[i:i + 8192]) for i in range(0, 57344, 8192)])
    o = O.TwoStageFFTConvolver("orc"); assert o.init(8192, 16384, ir)
    yo = np.concatenate([o.process(x[i:i + 8192]) for i in range(0, 57344, 8192)])
    assert rel_rms(y, yo) <= TOL
    # above the largest partition: accepted, served with 16384-sample partitions, same samples
    assert c.init(32768, ir) is True and c.last_error == 0 and c.head_block == 16384
    o = O.FFTConvolver("orc"); assert o.init(32768, ir)
    yo = np.concatenate([o.process(x[:32768]), o.process(x[32768:])])
    assert rel_rms(np.concatenate([c.process(x[:32768]), c.process(x[32768:])]), yo) <= TOL
    t = reevr_amd.TwoStageFFTConvolver()
    assert t.init(32768, 65536, ir) is True and (t.head_block, t.tail_block) == (16384, 16384)
    o = O.TwoStageFFTConvolver("orc"); assert o.init(32768, 65536, ir)
    assert rel_rms(np.concatenate([t.process(x[:32768]), t.process(x[32768:])]),
                   np.concatenate([o.process(x[:32768]), o.process(x[32768:])])) <= TOL
    d = reevr_amd.FFTConvolver(fft_f64=True)
    assert d.init(16384, ir, max_len=60000) is True and d.head_block == 8192   # f64 transforms: partitions up to 8192
    o = O.FFTConvolver("orc"); assert o.init(16384, ir)
    assert rel_rms(d.process(x), o.process(x)) <= 1e-6


def test_semantics_on_gpu():
    ir = synth.synth_ir(3000, 1, 4)[0]
    x = synth.synth_input(4096, 3)
    c = reevr_amd.TwoStageFFTConvolver()
    assert np.all(c.process(x[:100]) == 0)               # before init
    assert c.init(64, 256, ir)
    y1 = c.process(x)
    assert c.process(x[:0]).size == 0                    # len 0
    c.clear()
    y2 = c.process(x)                                    # clear(): fresh history, same IR
    assert np.array_equal(y1, y2)
    c.reset()
    assert np.all(c.process(x[:50]) == 0)                # reset(): zeros until init
    assert c.init(64, 256, ir)
    assert np.array_equal(c.process(x), y1)              # deterministic
    assert c.init(64, 256, np.zeros(100, np.float32))    # re-init with an all-zero IR
    assert np.all(c.process(x[:300]) == 0)
    # head > tail is swapped (TwoStageFFTConvolver.cpp:100-104)
    a = reevr_amd.TwoStageFFTConvolver(); assert a.init(256, 64, ir)
    assert np.array_equal(a.process(x), y1)


def test_stereo_convolver_quad_and_force2():
    """StereoConvolver fan-out (src/dsp/StereoConvolver.cpp:22-42) incl. quad and force2Chans."""
    class Imp:
        pass
    irs = synth.synth_ir(30000, 4, 6)
    imp = Imp()
    imp.bufferLL, imp.bufferRR, imp.bufferLR, imp.bufferRL = irs
    imp.isQuad = True
    sc = reevr_amd.StereoConvolver()
    sc.prepare(480)                                      # non-power-of-two host block -> head 512
    assert sc.headBlockSize == 512 and sc.tailBlockSize == 8192
    sc.loadImpulse(imp)
    L, R = synth.synth_input(480 * 40, 0), synth.synth_input(480 * 40, 1)
    got = {k: [] for k in "LL RR LR RL".split()}
    for i in range(0, len(L), 480):
        sc.process(L[i:i + 480], R[i:i + 480], 480)
        for k in got:
            got[k].append(getattr(sc, "buffer" + k).copy())
    feeds = dict(LL=L, RR=R, LR=L, RL=R)
    for idx, k in enumerate("LL RR LR RL".split()):
        o = O.TwoStageFFTConvolver("orc"); assert o.init(512, 8192, irs[idx])
        want = np.concatenate([o.process(feeds[k][i:i + 480]) for i in range(0, len(L), 480)])
        assert rel_rms(np.concatenate(got[k]), want) <= TOL
    before = sc.bufferLR.copy()
    sc.process(L[:480], R[:480], 480, True)              # force2Chans leaves LR/RL untouched
    assert np.array_equal(before, sc.bufferLR)
    assert isinstance(sc.finishedLoading(), bool)


def test_many_channels_one_set_vs_oracle():
    """BASELINE configs[3]/[4] shape: many independent channels in ONE set (one launch per stage),
    each with its own IR (different lengths: the set pads the partition count) and input."""
    nch, frames = 16, 40000
    irs = [synth.synth_ir(9000 + 1500 * c, 1, 30 + c)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(frames, c) for c in range(nch)])
    s = reevr_amd.ConvolverSet(nch)
    assert s.init(512, 8192, irs, max_len=frames)
    big = s.process(x)
    s.clear()
    blk = np.concatenate([s.process(x[:, i:i + 512]) for i in range(0, 512 * 40, 512)], axis=1)
    for c in range(nch):
        o = O.TwoStageFFTConvolver("orc"); assert o.init(512, 8192, irs[c])
        want = o.process(x[c])
        assert rel_rms(big[c], want) <= TOL
        assert rel_rms(blk[c], want[:512 * 40]) <= TOL


def test_f64_mode_big_call_and_streaming():
    case = cases.SYNTH_CASES["cfg4_inst3_10s_b512"]
    irs = cases.make_ir(case["ir"])
    x = cases.make_input(case, 2)
    want = cases.run_synth_case(orc_factory, case)
    s = reevr_amd.ConvolverSet(2, fft_f64=True)
    assert s.init(512, 8192, list(irs), max_len=case["frames"])
    big = s.process(x)
    s.clear()
    n = 512 * 100
    blk = np.concatenate([s.process(x[:, i:i + 512]) for i in range(0, n, 512)], axis=1)
    for c in range(2):
        assert rel_rms(big[c], want[c]) <= 1e-6        # double transforms: reference-grade
        assert rel_rms(blk[c], want[c, :n]) <= 1e-6


def test_handles_are_independent_across_threads():
    """The reference's threading contract: init() on one instance runs concurrently
    with process() on another (IR hot-swap, src/PluginProcessor.cpp:1680-1691)."""
    import threading
    ir_a = synth.synth_ir(60000, 2, 40)
    ir_b = synth.synth_ir(90000, 2, 41)
    x = np.stack([synth.synth_input(512 * 300, c) for c in range(2)])
    want = []
    for c in range(2):
        o = O.TwoStageFFTConvolver("orc"); assert o.init(512, 8192, ir_a[c])
        want.append(o.process(x[c]))
    want = np.stack(want)
    a = reevr_amd.ConvolverSet(2, bg_stream=True)
    assert a.init(512, 8192, list(ir_a))
    stop = threading.Event()
    errors = []

    def loader():          # keeps re-initialising and exercising a second set
        b = reevr_amd.ConvolverSet(2, bg_stream=True)
        try:
            while not stop.is_set():
                if not b.init(512, 8192, list(ir_b)):
                    errors.append(b.last_error_string)
                    return
                b.process(x[:, :512 * 4])
        except Exception as e:   # pragma: no cover
            errors.append(repr(e))

    t = threading.Thread(target=loader)
    t.start()
    try:
        got = np.concatenate([a.process(x[:, i:i + 512]) for i in range(0, x.shape[1], 512)], axis=1)
    finally:
        stop.set()
        t.join()
    assert not errors, errors
    for c in range(2):
        assert rel_rms(got[c], want[c]) <= TOL


def test_knobs_are_per_handle_across_threads():
    """rvc_set_create_tuned: a set's measurement knobs are its own. Two threads create, initialise and run sets with DIFFERENT
    knobs at the same time (one forces two levels of 32-block tiles and two child sets, the other one level and none) while
    the main thread flips the process-wide defaults: every set must report the plan of its own knobs and match the oracle."""
    import threading
    import torch
    head, tail, nch, nblk = 64, 256, 8, 16 * 12
    irs = [synth.synth_ir(2 * tail + 40 * tail - 37 * c, 1, 500 + c)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 90 + c) for c in range(nch)])
    want = []
    for c in range(nch):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        want.append(o.process(x[c]))
    errors, stop = [], threading.Event()

    def worker(tune, tiling, expect_tile, expect_subsets):
        try:
            dx = torch.from_numpy(x).cuda()
            for _ in range(6):
                s = reevr_amd.ConvolverSet(nch, time_tiling=tiling, tune=tune)
                assert s.init(head, tail, irs, max_len=head), s.last_error_string
                p = s.plan()
                assert (p["tail_tile_blocks"], p["subsets"]) == (expect_tile, expect_subsets), (tune, p)
                got = s.process_device_blocks(dx, head).cpu().numpy()
                assert s.last_error == 0, s.last_error_string
                s.close()
                for c in (0, nch - 1):
                    assert rel_rms(got[c], want[c]) <= TOL, (tune, c)
        except Exception as e:
            errors.append((tune, repr(e)))

    ts = [threading.Thread(target=worker, args=({"k1": 32, "subsets": 2}, "force2", 32, 2)),
          threading.Thread(target=worker, args=({"k1": 8, "two_level_min_p": 1000, "subsets": 1}, "force", 8, 1))]
    for t in ts:
        t.start()
    try:
        while any(t.is_alive() for t in ts):          # the old failure mode: a debug knob set by one thread changes another's plan
            for v in (16, 0):
                reevr_amd.set_tuning("k1", v)
            for v in (4, -1):
                reevr_amd.set_tuning("subsets", v)
    finally:
        reevr_amd.set_tuning("k1", 0)
        reevr_amd.set_tuning("subsets", -1)
        for t in ts:
            t.join()
    assert not errors, errors


@pytest.mark.parametrize("ir_len", [60000, 150000])
@pytest.mark.parametrize("fixed", [False, True])
def test_mixed_long_and_short_calls_hand_state_over(fixed, ir_len):
    """Adaptive partitioning: a long call is served by one uniform delay line at the tail block
    size and leaves the head stage's state (delay-line history, tail-ring rows, pre-multiplied
    accumulator) stale; the next short call must rebuild it lazily. Any interleaving of long,
    block-sized and ragged calls has to give the reference's stream."""
    # ir_len 150000 (> 4 x 16384) also creates the wide stage: 70000 / 90001-frame calls go through it,
    # 8192*5-frame calls through the tail-size line, and each switch rebuilds the other's delay line
    ir = synth.synth_ir(ir_len, 2, 50)
    sched = [512] * 3 + [70000] + [512] * 40 + [37, 475, 512, 300] + [8192 * 5] + [100] + [512] * 20 + \
            [8192 * 4 + 5] + [8187] + [512] * 33 + [90001] + [511, 1] + [512] * 17 + [8192 * 6] + [66000] + [512] * 5
    total = sum(sched)
    x = np.stack([synth.synth_input(total, c) for c in range(2)])
    s = reevr_amd.ConvolverSet(2, fixed_partitions=fixed)
    assert s.init(512, 8192, list(ir), max_len=max(sched))
    got = []
    pos = 0
    for n in sched:
        got.append(s.process(x[:, pos:pos + n]))
        pos += n
    got = np.concatenate(got, axis=1)
    assert s.last_error == 0, s.last_error_string
    for c in range(2):
        o = O.TwoStageFFTConvolver("orc"); assert o.init(512, 8192, ir[c])
        want = o.process(x[c])
        assert rel_rms(got[c], want) <= TOL
        # and no region is worse than the rest (a stale-state bug would be local)
        for a in range(0, total - 4096, 4096):
            seg = slice(a, a + 4096)
            assert np.sqrt(np.mean((got[c, seg].astype(np.float64) - want[seg]) ** 2)) <= 5e-6


@pytest.mark.parametrize("ir_len,block", [(48000, 512), (150000, 256), (20000, 64)])
def test_single_stage_set_long_and_short_calls(ir_len, block):
    """The same for a single-stage (FFTConvolver) set: calls touching several 8192-sample blocks go
    through the long-call stage (whole IR at block 8192; the wide stage too for the 150000-sample IR),
    whose delay line nobody keeps current between long calls -- it is rebuilt from the time ring.
    (ir_len 20000 x block 64: IR too short for a long-call stage -- the plain path must be unchanged.)"""
    ir = synth.synth_ir(ir_len, 2, 51)
    sched = [block] * 3 + [70000] + [block] * 40 + [37, block - 37, block, 300] + [8192 * 5] + [100] + [block] * 20 + \
            [8192 * 4 + 5] + [8187] + [block] * 33 + [90001] + [block - 1, 1] + [block] * 17 + [8192 * 6] + [66000] + [block] * 5
    total = sum(sched)
    x = np.stack([synth.synth_input(total, c) for c in range(2)])
    s = reevr_amd.ConvolverSet(2)
    assert s.init_uniform(block, list(ir), max_len=max(sched))
    assert s.tail_block == 0 and s.partitions(1) == 0          # still reports a single-stage geometry
    got = []
    pos = 0
    for n in sched:
        got.append(s.process(x[:, pos:pos + n]))
        pos += n
    got = np.concatenate(got, axis=1)
    assert s.last_error == 0, s.last_error_string
    want = []
    for c in range(2):
        o = O.FFTConvolver("orc"); assert o.init(block, ir[c])
        want.append(o.process(x[c]))
        assert rel_rms(got[c], want[c]) <= TOL
        for a in range(0, total - 4096, 4096):
            seg = slice(a, a + 4096)
            assert np.sqrt(np.mean((got[c, seg].astype(np.float64) - want[c][seg]) ** 2)) <= 5e-6
    # clear() and a different interleaving on the same set: long call first, then blocks, then long again
    s.clear()
    sched2 = [8192 * 7 + 3] + [block] * 9 + [50000] + [block] * 4
    got2 = []
    pos = 0
    for n in sched2:
        got2.append(s.process(x[:, pos:pos + n]))
        pos += n
    got2 = np.concatenate(got2, axis=1)
    for c in range(2):
        assert rel_rms(got2[c], want[c][:pos]) <= TOL


def test_device_entry_zeros_before_init_and_with_empty_ir():
    """process before init / with an all-zero IR writes zeros (FFTConvolver.cpp:157-161) -- also
    through the device-pointer entry, which has no stream yet at that point."""
    import torch
    x = torch.ones(2, 700, device="cuda")
    y = torch.full((2, 700), 3.0, device="cuda")
    s = reevr_amd.ConvolverSet(2)
    s.process_device(x, y)
    torch.cuda.synchronize()
    assert float(y.abs().max()) == 0.0
    y.fill_(3.0)
    assert s.init(64, 256, [np.zeros(10, np.float32)] * 2)
    s.process_device(x, y)
    torch.cuda.synchronize()
    assert float(y.abs().max()) == 0.0


def _random_schedule(rng, total, head, tail):
    """Call sizes mixing the regimes the engine treats differently: sub-block, exactly one block,
    a few blocks, and long calls that take the adaptive whole-IR path."""
    out, done = [], 0
    while done < total:
        kind = rng.randint(0, 6)
        if kind == 0:
            n = rng.randint(1, head + 1)
        elif kind == 1:
            n = head
        elif kind == 2:
            n = rng.randint(head, 4 * head + 1)
        elif kind == 3:
            n = rng.randint(1, 3 * tail)
        elif kind == 4:
            n = rng.randint(4 * tail, 9 * tail)
        else:
            n = head - (done % head) if done % head else head     # realign to the block grid
        n = max(1, min(n, total - done))
        out.append(n)
        done += n
    return out


@pytest.mark.parametrize("tiling", ["default", "force", "force2", "widen", "shrink"])
@pytest.mark.parametrize("seed", list(range(48)))
def test_fuzz_geometry_and_call_pattern(seed, tiling):
    """Seeded fuzz: random head/tail sizes (incl. non powers of two), IR lengths around the
    stage boundaries, 1-3 channels of different lengths, flags, and call patterns; every run is
    compared with the oracle sample by sample. tiling = force: the causal time tiling of the
    block-synchronous delay lines (RVC_FLAG_FORCE_TIME_TILING) whatever the stage size; force2: with two-level
    tiles (RVC_FLAG_FORCE_TWO_LEVEL: first-level sweeps of 16 blocks, second-level sweeps every 8). widen / shrink: the
    delay-1 tail stage of many-channel sets forced on these small ones (tail_slack = 1: the tail at twice the block,
    2: half the zero-latency stage; float32 transforms, tail job on the set's own stream, forced tiling)."""
    rng = np.random.RandomState(1000 + seed)
    slack = {"widen": 1, "shrink": 2}.get(tiling, -1)
    head = int(rng.choice([1, 3, 8, 24, 64, 100, 256, 512, 1024]))
    tail = int(rng.choice([max(head, 16), 2 * max(head, 8), 128, 512, 2048, 8192]))
    if head > tail:
        head, tail = tail, head
    hb = 1 << (max(head, 1) - 1).bit_length()
    tb = 1 << (max(tail, 1) - 1).bit_length()
    nch = int(rng.randint(1, 4))
    base = int(rng.choice([tb // 2 + 1, tb, tb + 1, 2 * tb - 1, 2 * tb, 2 * tb + 1, 3 * tb + 7, 7 * tb + 13, 12 * tb]))
    base = max(1, min(base, 120000))
    irs = []
    for c in range(nch):
        n = max(1, base - int(rng.randint(0, max(2, base // 3))) if c else base)
        irs.append(synth.synth_ir(n, 1, 60 + 3 * seed + c)[0])
    total = int(min(max(14 * tb, 4000), 200000))
    sched = _random_schedule(rng, total, hb, tb)
    bg = bool(rng.randint(0, 2))
    fixed = bool(rng.randint(0, 2))
    x = np.stack([synth.synth_input(total, 5 * seed + c) for c in range(nch)])
    if slack > 0:
        bg = fixed = False
        with reevr_amd.tuning(tail_slack=slack):
            s = reevr_amd.ConvolverSet(nch, bg_stream=False, time_tiling="force", fft_f32=True)
            assert s.init(head, tail, irs, max_len=max(sched)), s.last_error_string
        if s.partitions(1) > 0 and hb < tb:                         # a tail stage exists
            if slack == 2:
                assert s.tail_block == tb and s.partitions(0) == tb // hb
            elif 64 <= tb <= 8192:
                assert s.tail_block == 2 * tb and s.partitions(0) == 2 * tb // hb
    else:
        s = reevr_amd.ConvolverSet(nch, bg_stream=bg, fixed_partitions=fixed, time_tiling=True if tiling == "default" else tiling)
        assert s.init(head, tail, irs, max_len=max(sched)), s.last_error_string
    clear_at = int(rng.randint(0, len(sched))) if rng.randint(0, 3) == 0 else -1
    got = np.empty_like(x)
    pos = 0
    start = 0
    for i, n in enumerate(sched):
        if i == clear_at and pos % hb == 0:      # block-aligned clear (the mid-block quirk is documented)
            s.clear()
            start = pos
        got[:, pos:pos + n] = s.process(x[:, pos:pos + n])
        pos += n
    assert s.last_error == 0, s.last_error_string
    for c in range(nch):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        want = np.empty(total, np.float32)
        want[:start] = o.process(x[c, :start]) if start else want[:0]
        if start:
            o.clear()
        want[start:] = o.process(x[c, start:])
        err = np.sqrt(np.mean((got[c].astype(np.float64) - want) ** 2))
        ref = max(np.sqrt(np.mean(want.astype(np.float64) ** 2)), 1e-12)
        assert err / ref <= TOL, (f"seed {seed}: head {head} tail {tail} nch {nch} ir {[len(i) for i in irs]} "
                                  f"bg {bg} fixed {fixed} clear {clear_at}: rel rms {err / ref:.3e}")


@pytest.mark.parametrize("tiling", [True, False, "force", "force2", "force2_k32", "widen", "shrink_force2",
                                    "spread1_force2_k32", "spread3_force2", "spread3_shrink_force2", "spread3_widen", "spread2_force2_k32",
                                    "phases_force2_k32", "phases_force2", "phases_shrink_force2", "phases_widen", "phases_force",
                                    "phspread_force2_k32", "phspread_shrink_force2", "phspread_widen",
                                    "third_force", "third_force2_k32", "third_phases_force2_k32", "third_phases_shrink_force2", "third_phases_widen",
                                    "third_phases_force", "third_spread3_force2", "third_phspread_force2_k32",
                                    "f64_force", "f64_force2_k32", "f64_third_force", "f64_third_force2_k32", "f64_third_phases_force2_k32"])
@pytest.mark.parametrize("seed", list(range(24)) + [226])   # 226: 3 tail partitions on 4 sweep waves (a wave without work)
def test_fuzz_block_synchronous_time_tiling(seed, tiling):
    """The plug-in's calling pattern -- one call per host block, now and then a ragged one, several calls inside one
    block, a multi-block call or a block-aligned clear() -- over many sweep tiles (8 blocks each), with the causal
    time tiling off, on by size (default) and forced: every output sample against the oracle.
    spreadN_*: the tail stage's sweeps issued a tail period early in channel slices behind the per-block calls (knob
    tail_spread = N: bit 0 first-level, bit 1 second-level sweeps; what sets of >= 256 channels run by default) on top of the
    named tiling mode -- planned sweeps meeting ragged calls, multi-block calls (which drop them) and clear().
    third_*: third-level sweeps half way through every group of 8 tail blocks (knob tail_third; what sets of >= 256 channels run by
    default) on top of the named mode."""
    spread = phases = 0
    third = -1
    f64 = False
    if str(tiling).startswith("f64_"):           # every transform in double: every call takes the GENERAL path, whose zero-latency stage is
        f64, tiling = True, tiling[4:]           # time-tiled too (round 6: head_stage, rvc_set::head_gen)
    if str(tiling).startswith("third_"):
        third, tiling = 1, tiling[6:]
    if str(tiling).startswith("spread"):
        spread, tiling = int(tiling[6]), tiling[8:]
    elif str(tiling).startswith("phases_"):      # the tail tiles in channel groups out of phase (tail_phases; one group per channel here)
        phases, tiling = 8, tiling[7:]
    elif str(tiling).startswith("phspread_"):    # ... with every group's first-level sweep spread over the calls of the period before
        phases, spread, tiling = 8, 1, tiling[9:]
    rng = np.random.RandomState(4200 + seed)
    head = int(rng.choice([64, 128, 256, 512]))
    tail = int(rng.choice([2 * head, 4 * head, 16 * head]))
    nch = int(rng.randint(1, 4))
    n_tail_parts = int(rng.choice([1, 3, 9, 20]))
    base = 2 * tail + n_tail_parts * tail - int(rng.randint(0, tail // 2))
    irs = [synth.synth_ir(max(1, base - c * int(rng.randint(0, tail))), 1, 300 + 3 * seed + c)[0] for c in range(nch)]
    total = int(min(max(60 * tail, 40 * 8 * head), 300000))
    total -= total % head
    sched, done = [], 0
    while done < total:
        r = rng.randint(0, 40)
        if r == 0:
            n = int(rng.randint(1, head))                       # ragged: leaves the block grid ...
        elif r == 1 and done % head:
            n = head - done % head                              # ... and comes back to it
        elif r == 2:
            n = int(rng.randint(2, 6)) * head                   # a multi-block call (general path, drops the tiles)
        elif r == 3:
            n = int(rng.randint(5, 9)) * tail                   # a long call (adaptive path)
        else:
            n = head if done % head == 0 else head - done % head
        n = max(1, min(n, total - done))
        sched.append(n)
        done += n
    bg = bool(rng.randint(0, 2))
    x = np.stack([synth.synth_input(total, 11 * seed + c) for c in range(nch)])
    slack = {"widen": 1, "shrink_force2": 2}.get(tiling, -1)     # the delay-1 tail stage of many-channel sets, forced
    if slack > 0:
        bg = False
        tiling = "force" if slack == 1 else "force2"
    if spread:
        bg = False                                               # (a tail job on the second stream is never spread)
    with reevr_amd.tuning(k1=32 if tiling == "force2_k32" else 0, tail_slack=slack, tail_spread=spread, tail_phases=phases or -1, tail_third=third, head_third=third):   # (k1 = 32: first-level tiles of 32 blocks)
        s = reevr_amd.ConvolverSet(nch, bg_stream=bg, time_tiling="force2" if tiling == "force2_k32" else tiling, fft_f32=slack > 0, fft_f64=f64)
        assert s.init(head, tail, irs, max_len=max(sched)), s.last_error_string
    assert s.plan()["tail_third_level"] == (1 if third > 0 and s.tile_rows(1) else 0)
    if f64:
        assert s.plan()["block_path"] == 1 and s.plan()["head_patch_in_launch"] == 0
        assert (s.tile_rows(0) > 0) == (s.partitions(0) >= 3) and s.plan()["head_third_level"] == (1 if third > 0 and s.tile_rows(0) else 0)
    else:
        assert s.plan()["head_third_level"] == (1 if third > 0 and s.plan()["head_patch_in_launch"] else 0)
    if spread and s.tile_rows(1):
        assert s.plan()["tail_spread"] == ((spread if s.tile_rows(1) > 8 else spread & 1) if not phases else spread & 1)
    if phases and s.tile_rows(1):
        assert s.plan()["tail_phase_groups"] == min(nch, 8)
    if slack == 1:
        assert s.tail_block == 2 * tail and s.partitions(0) == 2 * tail // head
    elif slack == 2:
        assert s.tail_block == tail and s.partitions(0) == tail // head
    if str(tiling).startswith("force2"):
        assert s.tile_rows(1) == (32 if tiling == "force2_k32" else 16)
    clear_at = int(rng.randint(len(sched) // 4, len(sched))) if rng.randint(0, 3) == 0 else -1
    got = np.empty_like(x)
    pos = start = 0
    for i, n in enumerate(sched):
        if i >= clear_at >= 0 and pos % head == 0 and start == 0:
            s.clear()
            start = pos
        got[:, pos:pos + n] = s.process(x[:, pos:pos + n])
        pos += n
    assert s.last_error == 0, s.last_error_string
    for c in range(nch):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        want = np.empty(total, np.float32)
        want[:start] = o.process(x[c, :start]) if start else want[:0]
        if start:
            o.clear()
        want[start:] = o.process(x[c, start:])
        err = np.sqrt(np.mean((got[c].astype(np.float64) - want) ** 2))
        ref = max(np.sqrt(np.mean(want.astype(np.float64) ** 2)), 1e-12)
        assert err / ref <= TOL, (f"seed {seed}: head {head} tail {tail} nch {nch} ir {[len(i) for i in irs]} bg {bg} "
                                  f"tiling {tiling} clear@{start}: rel rms {err / ref:.3e}")


@pytest.mark.parametrize("seed", list(range(10)))
def test_fuzz_child_sets_call_patterns(seed):
    """The same call patterns with the set served by TWO CHILD SETS (forced: the default only does that from 2048 channels on):
    per-block calls through host pointers and through the device entry (per call and as the C block loop, torch's stream as
    producer and consumer, nothing synchronised on the host), ragged and multi-block calls, a block-aligned clear() -- every
    output sample of every channel against the oracle and against the same set without children."""
    import torch
    rng = np.random.RandomState(9100 + seed)
    head = int(rng.choice([64, 128, 256]))
    tail = int(rng.choice([2 * head, 4 * head, 16 * head]))
    nch = int(rng.choice([4, 6, 8]))
    n_tail_parts = int(rng.choice([1, 3, 9, 20]))
    base = 2 * tail + n_tail_parts * tail - int(rng.randint(0, tail // 2))
    irs = [synth.synth_ir(max(1, base - c * int(rng.randint(0, tail))), 1, 600 + 3 * seed + c)[0] for c in range(nch)]
    total = int(min(max(40 * tail, 30 * 8 * head), 200000))
    total -= total % head
    sched, done = [], 0
    while done < total:
        r = rng.randint(0, 30)
        if r == 0:
            n, how = int(rng.randint(1, head)), "host"                        # ragged ...
        elif r == 1 and done % head:
            n, how = head - done % head, "host"                                # ... and back to the grid
        elif r == 2:
            n, how = int(rng.randint(2, 5)) * head, "device"                   # a multi-block call
        elif r == 3 and done % head == 0:
            n, how = int(rng.randint(3, 12)) * head, "loop"                    # the C block loop over several blocks
        else:
            n = head if done % head == 0 else head - done % head
            how = "device" if rng.randint(0, 2) else "host"
        n = max(1, min(n, total - done))
        sched.append((n, how))
        done += n
    bg = bool(rng.randint(0, 2))
    tiling = ["force", "force2", True][seed % 3]
    # every other seed: spread tail sweeps (tail_spread 1 / 3) with the children's tiles out of phase (kid_stagger)
    spread = [0, 3, 0, 1][seed % 4]
    phases = [0, 0, 4, 0][seed % 4]           # seeds 2, 6: the tail tiles of each child in channel groups out of phase
    if spread:
        bg = False
    x = np.stack([synth.synth_input(total, 13 * seed + c) for c in range(nch)])
    dx = torch.from_numpy(x).cuda()
    clear_at = int(rng.randint(len(sched) // 4, len(sched))) if rng.randint(0, 2) == 0 else -1
    outs, start = [], 0
    for kids in (2, 1):
        with reevr_amd.tuning(subsets=kids, tail_spread=spread, kid_stagger=1 if spread else -1, tail_phases=phases or -1):
            s = reevr_amd.ConvolverSet(nch, bg_stream=bg, time_tiling=tiling)
            assert s.init(head, tail, irs, max_len=max(n for n, _ in sched)), s.last_error_string
        assert s.subsets == kids
        got = torch.zeros_like(dx)
        pos = start = 0
        for i, (n, how) in enumerate(sched):
            if i >= clear_at >= 0 and pos % head == 0 and start == 0:
                s.sync()
                s.clear()
                start = pos
            if how == "host":
                torch.cuda.synchronize()
                got[:, pos:pos + n] = torch.from_numpy(s.process(x[:, pos:pos + n])).cuda()
            elif how == "loop":
                s.process_device_blocks(dx[:, pos:pos + n], head, got[:, pos:pos + n], sync=False)
            else:
                s.process_device(dx[:, pos:pos + n], got[:, pos:pos + n], sync=False)
            pos += n
        s.sync()
        torch.cuda.synchronize()
        assert s.last_error == 0, s.last_error_string
        outs.append(got.cpu().numpy())
        s.close()
    # (Same samples up to the last bit or two: a child whose channels all carry shorter impulses than the set's longest has
    #  fewer partitions in its delay lines, and the partition-split sweeps then associate their partial sums differently --
    #  bit identity for equal partition counts is test_child_sets_match_single_set's claim.)
    for c in range(nch):
        assert rel_rms(outs[0][c], outs[1][c]) <= 1e-6, f"seed {seed}: child sets differ from the single set, channel {c}"
    for c in range(nch):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        want = np.empty(total, np.float32)
        want[:start] = o.process(x[c, :start]) if start else want[:0]
        if start:
            o.clear()
        want[start:] = o.process(x[c, start:])
        assert rel_rms(outs[0][c], want) <= TOL, (f"seed {seed}: head {head} tail {tail} nch {nch} bg {bg} tiling {tiling} "
                                                  f"clear@{start}: {rel_rms(outs[0][c], want):.3e}")


def test_time_tiling_many_channels_device_blocks():
    """64 lock-step channels (the size at which the zero-latency stage tiles by default), block-synchronous through the
    device entry, tiling on vs off vs the oracle on three channels."""
    import torch
    nch, head, tail, ir_len, nblk = 64, 512, 8192, 70000, 200
    irs = [synth.synth_ir(ir_len, 1, 500 + c)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 40 + c) for c in range(nch)])
    dx = torch.from_numpy(x).cuda()
    outs = {}
    for tiling in (True, False):
        s = reevr_amd.ConvolverSet(nch, time_tiling=tiling)
        assert s.init(head, tail, irs, max_len=head)
        outs[tiling] = s.process_device_blocks(dx, head).cpu().numpy()
        assert s.last_error == 0
        s.close()
    for c in range(nch):
        assert rel_rms(outs[True][c], outs[False][c]) <= 2e-6, c
    for c in (0, 31, 63):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        assert rel_rms(outs[True][c], o.process(x[c])) <= TOL, c


# (test_lockstep_1024_channels_at_bench_geometry / test_lockstep_4096_channels_impulse_identity of rounds 2-5 stopped before the
#  delay lines were full; their steady-state successors at bench.py's own plans live in tests/test_gpu_steady_state.py.)


@pytest.mark.parametrize("tiling", [False, True, "force", "force2"])
@pytest.mark.parametrize("head,tail,parts,nch", [(64, 128, 5, 3), (64, 1024, 3, 2), (256, 512, 12, 2), (512, 8192, 2, 4),
                                                  (128, 2048, 20, 1)])
def test_device_block_loop_tail_on_second_stream(head, tail, parts, nch, tiling):
    """The asynchronous per-block loop (rvc_set_process_device_blocks, nothing waits on the host) with the tail stage on
    the second stream: the foreground stream runs a whole tail period ahead of each tail job (Convolver.cpp:84-95 /
    TwoStageFFTConvolver.cpp:213-222 hook points), so every ring the two streams share is exercised with both of them
    busy. Short tail periods (2 .. 16 head blocks), many of them; every sample against the oracle; twice, to catch
    anything that depends on timing."""
    import torch
    ir_len = 2 * tail + parts * tail - tail // 3
    irs = [synth.synth_ir(ir_len - 17 * c, 1, 700 + c)[0] for c in range(nch)]
    nblk = max(40 * (tail // head), 600)
    x = np.stack([synth.synth_input(head * nblk, 70 + c) for c in range(nch)])
    dx = torch.from_numpy(x).cuda()
    want = []
    for c in range(nch):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        want.append(o.process(x[c]))
    for rep in range(2):
        s = reevr_amd.ConvolverSet(nch, bg_stream=True, time_tiling=tiling)
        assert s.init(head, tail, irs, max_len=head)
        got = s.process_device_blocks(dx, head).cpu().numpy()
        assert s.last_error == 0, s.last_error_string
        s.close()
        for c in range(nch):
            assert rel_rms(got[c], want[c]) <= TOL, (rep, c, rel_rms(got[c], want[c]))


@pytest.mark.parametrize("seed", list(range(16)))
def test_fuzz_single_stage_sets(seed):
    """The same for single-stage (FFTConvolver) sets: block sizes 16..4096, IR lengths on either side of
    the long-call stage's threshold (2 x 8192) and of the wide stage's (4 x 16384), call patterns
    mixing per-block, ragged and very long calls, optional block-aligned clear."""
    rng = np.random.RandomState(7000 + seed)
    block = int(rng.choice([16, 100, 256, 512, 2048, 4096]))
    hb = 1 << (block - 1).bit_length()
    nch = int(rng.randint(1, 3))
    base = int(rng.choice([900, 16384, 16385, 30000, 65536, 65537, 100000]))
    irs = [synth.synth_ir(max(1, base - (c * 777)), 1, 160 + 3 * seed + c)[0] for c in range(nch)]
    total = 260000
    sched = []
    left = total
    while left > 0:
        kind = rng.randint(0, 6)
        n = hb if kind < 3 else (int(rng.randint(1, 3 * hb + 2)) if kind == 3 else int(rng.choice([8192 * 4 + 1, 8192 * 5, 70000, 16384 * 4 + 3])))
        n = min(n, left)
        sched.append(n)
        left -= n
    x = np.stack([synth.synth_input(total, 9 * seed + c) for c in range(nch)])
    s = reevr_amd.ConvolverSet(nch, fixed_partitions=bool(rng.randint(0, 4) == 0))
    assert s.init_uniform(block, irs, max_len=max(sched)), s.last_error_string
    clear_at = int(rng.randint(0, len(sched))) if rng.randint(0, 2) == 0 else -1
    got = np.empty_like(x)
    pos = start = 0
    for i, n in enumerate(sched):
        if i == clear_at and pos % hb == 0:
            s.clear()
            start = pos
        got[:, pos:pos + n] = s.process(x[:, pos:pos + n])
        pos += n
    assert s.last_error == 0, s.last_error_string
    for c in range(nch):
        o = O.FFTConvolver("orc")
        assert o.init(block, irs[c])
        want = np.empty(total, np.float32)
        want[:start] = o.process(x[c, :start]) if start else want[:0]
        if start:
            o.clear()
        want[start:] = o.process(x[c, start:])
        err = np.sqrt(np.mean((got[c].astype(np.float64) - want) ** 2))
        ref = max(np.sqrt(np.mean(want.astype(np.float64) ** 2)), 1e-12)
        assert err / ref <= TOL, f"seed {seed}: block {block} nch {nch} ir {[len(i) for i in irs]} clear {clear_at}: {err / ref:.3e}"


def test_ir_hot_swap_matches_reference_sequence():
    """load -> warm-up replay (one multi-block call here, a loop of block calls
    in the reference) -> 50 ms crossfade -> swap, for a quad impulse, against the oracle-side
    restatement of src/PluginProcessor.cpp:1655-1756, 1793-1838."""
    from reevr_amd.hotswap import HotSwapStereoConvolver
    from tests.ref_hotswap import RefHotSwap

    class Imp:
        pass

    def imp(inst, n, quad):
        irs = synth.synth_ir(n, 4, inst)
        m = Imp()
        m.bufferLL, m.bufferRR, m.bufferLR, m.bufferRL = irs
        m.isQuad = quad
        return m

    sr, blk, nblocks = 48000, 480, 70
    a, b = imp(70, 30000, True), imp(71, 22000, True)
    L = synth.synth_input(blk * nblocks, 0); R = synth.synth_input(blk * nblocks, 1)
    gpu = HotSwapStereoConvolver(lambda: reevr_amd.StereoConvolver(), threaded=True)
    ref = RefHotSwap()
    for h in (gpu, ref):
        h.prepare(sr, blk)
        h.loadImpulse(a)
    got, want = [], []
    for i in range(nblocks):
        s = slice(i * blk, (i + 1) * blk)
        if i == 30:
            assert gpu.request_impulse(b)
            gpu.wait_loaded()                 # deterministic test: the load finishes before the next block
            ref.request_impulse(b)
        got.append(gpu.process(L[s], R[s], L[s], R[s], blk))
        want.append(ref.process(L[s], R[s], L[s], R[s], blk))
    got = np.concatenate(got, axis=1); want = np.concatenate(want, axis=1)
    assert gpu.loadState == 0 and ref.state == 0          # faded and swapped
    for c in range(2):
        assert rel_rms(got[c], want[c]) <= TOL
    fade = slice(31 * blk, 38 * blk)                      # the crossfade region itself
    assert np.sqrt(np.mean((got[:, fade].astype(np.float64) - want[:, fade]) ** 2)) <= 5e-6


def test_reinit_same_geometry_keeps_buffers_and_is_exact():
    """IR hot-swap: init() with a new IR of unchanged geometry re-uses all device state (fast path);
    with a different partition count it rebuilds. Either way the result is that of a fresh object."""
    x = synth.synth_input(512 * 60, 0)
    ir_a = synth.synth_ir(50000, 1, 80)[0]
    ir_b = synth.synth_ir(50000, 1, 81)[0]          # same length -> same partition counts
    ir_c = synth.synth_ir(23000, 1, 82)[0]          # fewer tail partitions -> full re-init
    s = reevr_amd.TwoStageFFTConvolver()
    assert s.init(512, 8192, ir_a)
    s.process(x[:512 * 37 + 100])                   # leave it mid-block with history
    for ir in (ir_b, ir_c, ir_a):
        assert s.init(512, 8192, ir)
        got = np.concatenate([s.process(x[i:i + 512]) for i in range(0, len(x), 512)])
        o = O.TwoStageFFTConvolver("orc"); assert o.init(512, 8192, ir)
        assert rel_rms(got, o.process(x)) <= TOL


def test_device_entry_with_misaligned_views_and_strides():
    """rvc_set_process_device on tensor VIEWS: pointers that are only 4-byte aligned and row strides
    larger than the call (the vectorised 8/16-byte paths must fall back, not misread)."""
    import torch
    ir = synth.synth_ir(40000, 2, 95)
    frames = 8192 * 6 + 37
    x = np.stack([synth.synth_input(frames, c) for c in range(2)])
    want = []
    for c in range(2):
        o = O.TwoStageFFTConvolver("orc"); assert o.init(512, 8192, ir[c])
        want.append(o.process(x[c]))
    want = np.stack(want)
    for off_in, off_out in ((1, 3), (2, 1), (0, 0), (3, 2)):
        big_in = torch.zeros(2, frames + 8, device="cuda")
        big_out = torch.zeros(2, frames + 8, device="cuda")
        big_in[:, off_in:off_in + frames] = torch.from_numpy(x).cuda()
        s = reevr_amd.ConvolverSet(2)
        assert s.init(512, 8192, list(ir), max_len=frames)
        # one long call (adaptive path, input read in place), then the same stream in awkward pieces
        s.process_device(big_in[:, off_in:off_in + frames], big_out[:, off_out:off_out + frames])
        got = big_out[:, off_out:off_out + frames].cpu().numpy()
        for c in range(2):
            assert rel_rms(got[c], want[c]) <= TOL
        s.clear()
        pos = 0
        for n in (511, 513, 8191, 20000, frames - 29215):
            s.process_device(big_in[:, off_in + pos:off_in + pos + n], big_out[:, off_out + pos:off_out + pos + n])
            pos += n
        assert pos == frames
        got = big_out[:, off_out:off_out + frames].cpu().numpy()
        for c in range(2):
            assert rel_rms(got[c], want[c]) <= TOL


@pytest.mark.parametrize("case", ["f64_512", "head2048"])
def test_general_per_block_path_time_tiled_vs_oracle(case):
    """Round 6: the zero-latency stage of sets whose per-block call takes the GENERAL path (forward transform / delay line / inverse
    launches) is time-tiled as well -- by size, no flag: (f64_512) 256 channels of BASELINE config 2's geometry with every transform in
    double (RVC_FLAG_FFT_F64: the mode at the reference's own precision; 16 head partitions: tiles of 8 with third-level sweeps);
    (head2048) 512 channels, head 2048 / tail 16384 (many channels with a large head block; 8 head partitions). One process() per head
    block through the device entry for 12 tail periods; the timed delay-line family must be patch launches with sweeps beside them, and
    five channels must equal the oracle."""
    import torch
    if case == "f64_512":
        nch, head, tail, kw = 256, 512, 8192, dict(fft_f64=True)
    else:
        nch, head, tail, kw = 512, 2048, 16384, {}
    nblk = 12 * (tail // head)
    irs = [synth.synth_ir(2 * tail + 5 * tail - 211 * (c % 6), 1, 30 + c % 5)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 700 + c % 7) for c in range(nch)])
    s = reevr_amd.ConvolverSet(nch, timing=True, **kw)
    assert s.init(head, tail, irs, max_len=head), s.last_error_string
    pl = s.plan()
    assert pl["block_path"] == 1 and pl["head_patch_in_launch"] == 0 and pl["head_tile_blocks"] == 8 and s.partitions(0) >= 8, pl
    got = s.process_device_blocks(torch.from_numpy(x).cuda(), head).cpu().numpy()
    assert s.last_error == 0, s.last_error_string
    n_fir, n_sweep, n_third = (s.kernel_time(k)[0] for k in (2, 9, 14))
    s.close()
    # one sweep per tile of 8 blocks, one third-level sweep per tile where the stage has them, a patch for every other block
    assert abs(n_sweep - nblk // 8) <= 1 and n_third == (nblk // 8 if pl["head_third_level"] else 0), (n_sweep, n_third, nblk)
    assert n_fir == nblk - n_sweep - n_third, (n_fir, n_sweep, n_third)
    assert np.isfinite(got).all()
    for c in (0, 1, nch // 2, nch - 2, nch - 1):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        assert rel_rms(got[c], o.process(x[c])) <= TOL, c


@pytest.mark.parametrize("phases", [1, 8])
def test_third_level_sweeps_run_where_planned_and_change_only_the_association(phases):
    """Round 6: third-level sweeps (Tile::s3; knobs tail_third / head_third) on a 16-channel set at BASELINE config 2's geometry in the
    delay-1 form (head 512 x 16, tail 8192 x 58: tail tiles of 32 blocks, head tiles of 8), block-synchronous for 70 tail blocks, with
    and without phase groups: (1) the launches happen where the schedule says -- one third-level head sweep per 8 head blocks, one
    third-level tail sweep per group of 8 tail blocks and phase group -- and the patch launches move fewer bytes with them than without
    (timed families 13 / 14 / 5); (2) the samples equal the two-level schedule's to float association (<= 2e-6) and the oracle's
    within the tolerance."""
    import torch
    nch, head, tail, nblk = 16, 512, 8192, 70 * 16
    irs = [synth.synth_ir(480000 - 997 * (c % 5), 1, 70 + c % 3)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 500 + c % 4) for c in range(nch)])
    dx = torch.from_numpy(x).cuda()
    outs, counts = {}, {}
    for third in (0, 1):
        s = reevr_amd.ConvolverSet(nch, timing=True, tune=dict(tail_slack=2, tail_phases=phases, tail_third=third, head_third=third, k1=32),
                                   time_tiling="force2", fft_f32=True)
        assert s.init(head, tail, irs, max_len=head), s.last_error_string
        pl = s.plan()
        assert (pl["tail_third_level"], pl["head_third_level"]) == (third, third) and pl["tail_phase_groups"] == phases
        assert pl["head_patch_in_launch"] == 1 and pl["tail_delay"] == 1 and (s.partitions(0), s.partitions(1)) == (16, 58)
        outs[third] = s.process_device_blocks(dx, head).cpu().numpy()
        assert s.last_error == 0, s.last_error_string
        counts[third] = {k: s.kernel_time(k)[0] for k in (5, 10, 12, 13, 14)}
        s.close()
    assert counts[0][13] == 0 and counts[0][14] == 0
    # head: tiles of 32 blocks (k1 = 32, forced two-level): every group of 8 has its half-way sweep
    assert abs(counts[1][14] - nblk // 8) <= 2, counts
    # tail: 70 tail blocks (the first is block 1: delay 1), one third-level sweep per group of 8 and phase group; a group's first tile is
    # up to 31 blocks short
    per_group = 70 // 8
    assert phases * (per_group - 5) <= counts[1][13] <= phases * (per_group + 1), counts
    # the patch launch: one per tail block in both schedules (phase groups: one for all groups)
    assert counts[1][5] <= counts[0][5]
    for c in range(nch):
        assert rel_rms(outs[1][c], outs[0][c]) <= 2e-6, c
    for c in (0, 7, 15):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        assert rel_rms(outs[1][c], o.process(x[c])) <= TOL, c


@pytest.mark.parametrize("slack", [0, 1, 2, "1_phases", "1_phases_third"])
def test_two_level_tiling_at_config3_geometry(slack):
    """(slack: what the tail's period of slack buys -- 0 the reference's structure, 1 the tail at block 16384 with delay 1 = what
    sets of >= 256 channels run at this geometry, 2 half the zero-latency stage; 1_phases: 1 with the tail tiles in 8 channel
    groups out of phase -- bench.py's config-3 plan but for the channel count.)
    BASELINE configs[2]'s geometry with the time tiling FORCED on both stages and many channels: head 256 / tail 8192,
    a 30 s @ 96 kHz IR on channel 0 (P_A = 64 zero-latency partitions, P_T = 350 tail partitions -> two-level tiles on
    both stages), 64 lock-step channels with IRs of different lengths, one process() per 256-frame block through the
    device entry until every tail partition carries signal; three channels against the oracle."""
    import torch
    nch, head, tail = 64, 256, 8192
    lens = [2880000] + [int(2880000 * (0.2 + 0.8 * ((7 * c) % nch) / nch)) for c in range(1, nch)]
    base = synth.synth_ir(2880000, 2, 0)
    irs = [base[c % 2][:lens[c]].copy() for c in range(nch)]
    nblk = (352 + 24) * (tail // head)                       # all 350 tail partitions in use, then a few tiles more
    x = np.stack([synth.synth_input(head * nblk, 200 + c % 5) for c in range(nch)])
    third = 1 if slack == "1_phases_third" else -1          # (round 6: + third-level sweeps on the 16384-bin rows, what the bench's set runs)
    phases = 8 if str(slack).startswith("1_phases") else -1
    slack = 1 if str(slack).startswith("1_phases") else slack
    with reevr_amd.tuning(tail_slack=slack, tail_phases=phases, tail_third=third):
        s = reevr_amd.ConvolverSet(nch, time_tiling="force")
        assert s.init(head, tail, irs, max_len=head), s.last_error_string
    assert (s.partitions(0), s.partitions(1), s.tail_block) == [(64, 350, 8192), (64, 175, 16384), (32, 351, 8192)][slack]
    assert s.plan()["tail_phase_groups"] == (8 if phases > 0 else 1) and s.plan()["tail_third_level"] == (1 if third > 0 else 0)
    assert s.tile_rows(0) > 8 and s.tile_rows(1) > 8         # two levels on both stages
    got = s.process_device_blocks(torch.from_numpy(x).cuda(), head).cpu().numpy()
    assert s.last_error == 0, s.last_error_string
    s.close()
    assert np.isfinite(got).all()
    for c in (0, 1, nch - 1):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        assert rel_rms(got[c], o.process(x[c])) <= TOL, c


def test_child_sets_flag():
    """Child sets are the default since round 4: 2048 block-synchronous channels run as two child sets, RVC_FLAG_NO_SUBSETS keeps
    one set on one queue -- same bits; sets below 2048 channels and long-call sets stay single."""
    import torch
    nch, head, nblk = 2048, 64, 24
    irs = [synth.synth_ir(300 + c % 17, 1, 700 + c % 29)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 70 + c % 11) for c in range(nch)])
    dx = torch.from_numpy(x).cuda()
    outs = []
    for flag, want in ((False, 1), (None, 2), ("unfenced", 2)):     # one queue / the default (fenced) / RVC_FLAG_CHILD_SETS
        s = reevr_amd.ConvolverSet(nch, child_sets=flag)
        assert s.init_uniform(head, irs, max_len=head), s.last_error_string
        assert s.subsets == want
        outs.append(s.process_device_blocks(dx, head).cpu().numpy())
        assert s.last_error == 0, s.last_error_string
        s.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    # The caller orders against ONE stream (rvc_set_stream(s, 0)) although the children run on their own: the producer of d_in
    # and the consumer of d_out live on torch's current stream, nothing synchronises on the host in between (sync=False), the
    # input buffer is overwritten right behind the call -- per-block calls (a fence per call) and the C block loop (one fence).
    # (with RVC_FLAG_CHILD_SETS -- no fences inside -- the Python mirror orders against every child's stream instead)
    for mode, form in ((None, "calls"), (None, "loop"), ("unfenced", "calls"), ("unfenced", "loop")):
        s = reevr_amd.ConvolverSet(nch, child_sets=mode)
        assert s.init_uniform(head, irs, max_len=head) and s.subsets == 2
        buf = torch.empty_like(dx)
        acc = torch.zeros_like(dx)
        big = torch.empty((64, 1 << 20), device="cuda")
        big.normal_()                                       # (keeps torch's stream busy: the copy below completes late)
        buf.copy_(dx, non_blocking=True)                    # producer of d_in, on torch's stream
        if form == "loop":
            y = s.process_device_blocks(buf, head, sync=False)
        else:
            y = torch.empty_like(buf)
            for b in range(nblk):
                s.process_device(buf[:, b * head:(b + 1) * head], y[:, b * head:(b + 1) * head], sync=False)
        acc += y                                            # consumer of d_out, on torch's stream
        buf.fill_(float("nan"))                             # the input is dead the moment the call's work is ordered
        torch.cuda.synchronize()
        assert np.array_equal(acc.cpu().numpy(), outs[0]), (mode, form)
        s.close()
    for c in (0, 1023, 1024, 2047):
        o = O.FFTConvolver("orc")
        assert o.init(head, irs[c])
        assert rel_rms(outs[1][c], o.process(x[c])) <= TOL, c
    # one handle through both structures: children -> single set (long calls) -> children again (the parent gives its streams up
    # when it gets children and creates them anew when it loses them)
    s = reevr_amd.ConvolverSet(nch)
    assert s.init_uniform(head, irs, max_len=head) and s.subsets == 2
    assert np.array_equal(s.process_device_blocks(dx, head).cpu().numpy(), outs[0])
    assert s.init_uniform(head, irs, max_len=nblk * head) and s.subsets == 1
    y = s.process_device(dx).cpu().numpy()
    for c in (0, 2047):
        assert rel_rms(y[c], outs[0][c]) <= 2e-6, c
    assert s.init_uniform(head, irs, max_len=head) and s.subsets == 2
    assert np.array_equal(s.process_device_blocks(dx, head).cpu().numpy(), outs[0])
    assert s.last_error == 0, s.last_error_string
    s.close()
    small = reevr_amd.ConvolverSet(64, child_sets=True)
    assert small.init_uniform(head, irs[:64], max_len=head) and small.subsets == 1
    small.close()
    long_calls = reevr_amd.ConvolverSet(nch, child_sets=True)
    assert long_calls.init_uniform(head, irs, max_len=64 * head) and long_calls.subsets == 1
    long_calls.close()


@pytest.mark.parametrize("kids", [2, 3, 4])
def test_child_sets_match_single_set(kids):
    """A set served by child sets on their own streams (rvc_set_subsets) gives, bit for bit, what the same set gives
    alone -- per-block device calls, a multi-block call, a host-pointer call, clear() -- and matches the oracle. Three children
    of eight channels: UNEVEN children (3 + 3 + 2; round 5: the remainder is dealt out one by one), partly filled workgroups
    (four channels per workgroup at head 128)."""
    import torch
    nch, head, tail, nblk = 8, 128, 512, 120
    irs = [synth.synth_ir(2 * tail + 5 * tail - 31 * c, 1, 900 + c)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 50 + c) for c in range(nch)])
    dx = torch.from_numpy(x).cuda()
    outs = []
    for n in (1, kids):
        reevr_amd.set_tuning("subsets", n)
        try:
            s = reevr_amd.ConvolverSet(nch, bg_stream=True, time_tiling="force")
            assert s.init(head, tail, irs, max_len=4 * head), s.last_error_string
        finally:
            reevr_amd.set_tuning("subsets", -1)
        assert s.subsets == n
        a = s.process_device_blocks(dx[:, :head * 100].contiguous(), head).cpu().numpy()
        b = s.process_device(dx[:, head * 100:head * 104].contiguous()).cpu().numpy()      # a multi-block call
        c = s.process(x[:, head * 104:head * 105])                                         # host pointers
        s.clear()
        d = s.process_device_blocks(dx[:, :head * 20].contiguous(), head).cpu().numpy()
        assert s.last_error == 0, s.last_error_string
        s.close()
        outs.append((a, b, c, d))
    for u, v in zip(*outs):
        assert np.array_equal(u, v)
    whole = np.concatenate(outs[1][:3], axis=1)
    for c in (0, nch - 1):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        assert rel_rms(whole[c], o.process(x[c, :head * 105])) <= TOL, c


def test_tail_slack_policy_and_children():
    """The delay-1 tail stage (rvc_plan.cpp plan_stages, rvc_state.cpp do_init): lock-step sets of >= 256 channels whose tail job runs on the set's own
    stream spend the tail period of slack the reference keeps for its background thread -- long tails (>= 128 partitions) run at
    TWICE the requested block, the others give half of the zero-latency stage to the tail; sets with the tail on a second
    stream, fixed partitions, the reference-order schedule or fewer channels keep the reference's structure; the children of a
    set decide TOGETHER (on the longest impulse of the whole set). Every variant against the oracle."""
    import torch
    head, tail, nblk = 64, 256, 700
    long_ir, short_ir = 2 * tail + 140 * tail - 17, 2 * tail + 20 * tail - 5
    x1 = np.stack([synth.synth_input(head * nblk, 70 + c) for c in range(4)])

    def run(nch, lens, expect, subsets=-1, reinit=False, **kw):
        irs = [synth.synth_ir(lens[c % len(lens)] - (c % 7), 1, 800 + c % 11)[0] for c in range(nch)]
        x = x1[np.arange(nch) % 4]
        plain = not kw and subsets < 0      # (a set created without flags or forced children: what reevr_amd.stage_plan describes)
        with reevr_amd.tuning(subsets=subsets):
            s = reevr_amd.ConvolverSet(nch, bg_stream=kw.pop("bg_stream", False), **kw)
            assert s.init(head, tail, irs, max_len=head), s.last_error_string
        assert (s.partitions(0), s.tail_block, s.partitions(1)) == expect, (kw, s.partitions(0), s.tail_block, s.partitions(1))
        if plain:                            # ... which is what the pure plan function (the CPU tests pin it) says for this request
            plan = reevr_amd.stage_plan(nch, head, tail, max(len(i) for i in irs))
            assert (plan["partitions"][0], plan["tail_block"], plan["partitions"][1]) == expect
        if subsets > 1:
            assert s.subsets == subsets
        got = s.process_device_blocks(torch.from_numpy(x).cuda(), head).cpu().numpy()
        assert s.last_error == 0, s.last_error_string
        got2 = None
        if reinit:                        # IR hot-swap with unchanged geometry (the fast path keeps every buffer), mid-stream
            irs2 = [(-0.5 * ir).astype(np.float32) for ir in irs]
            assert s.init(head, tail, irs2, max_len=head), s.last_error_string
            assert (s.partitions(0), s.tail_block, s.partitions(1)) == expect
            got2 = s.process_device_blocks(torch.from_numpy(x).cuda(), head).cpu().numpy()
            assert s.last_error == 0, s.last_error_string
        s.close()
        for c in (0, 1, nch - 1):
            o = O.TwoStageFFTConvolver("orc")
            assert o.init(head, tail, irs[c])
            assert rel_rms(got[c], o.process(x[c])) <= TOL, (kw, c)
            if got2 is not None:
                assert o.init(head, tail, irs2[c])
                assert rel_rms(got2[c], o.process(x[c])) <= TOL, (kw, c, "after re-init")

    run(256, [long_ir], (8, 512, 70), reinit=True)             # long tail: widened
    run(256, [short_ir], (4, 256, 21), reinit=True)            # short tail: the zero-latency stage shrinks
    run(255, [long_ir], (8, 256, 140))                         # below the channel threshold: the reference's structure
    run(256, [long_ir], (8, 256, 140), bg_stream=True)         # tail on the second stream: the slack is in use
    run(256, [long_ir], (8, 256, 140), fixed_partitions=True)
    run(256, [long_ir], (8, 256, 140), time_tiling=False)      # reference-order schedule
    run(256, [long_ir], (4, 256, 141), fft_f64=True)           # double transforms: no 2T-block transform, so it shrinks
    # two children: only channel 0 carries the long impulse -- the second child alone would shrink, together they widen
    run(512, [long_ir] + [short_ir] * 511, (8, 512, 70), subsets=2)
    # children plan with the WHOLE set's channel count (rvc_debug_plan's n_channels): 300 channels over two children of 150 run the
    # delay-1 plan of a 300-channel set, not the reference's structure a 150-channel set of its own would get
    run(300, [long_ir], (8, 512, 70), subsets=2)
    assert reevr_amd.stage_plan(300, head, tail, long_ir)["tail_delay"] == 1 and reevr_amd.stage_plan(150, head, tail, long_ir)["tail_delay"] == 2


def test_child_sets_with_empty_impulses_in_the_first_child():
    """A 4096-channel lock-step set is served by two children; when every impulse of the FIRST child's channels is empty that
    child holds no stages -- but the set's ordering stream (rvc_set_stream(s, 0)) and the fences of the other child are anchored
    on its stream, so it must still have one: the live half against the oracle, the empty half zeros (asynchronously, on that
    stream), with torch's producer / consumer ordered through stream 0 only."""
    import torch
    nch, head, tail, nblk = 4096, 512, 8192, 40
    live = synth.synth_ir(3 * tail + 999, 2, 7)
    irs = [np.zeros(0, np.float32)] * (nch // 2) + [live[c % 2][:3 * tail + 999 - 13 * (c % 5)].copy() for c in range(nch // 2)]
    x4 = np.stack([synth.synth_input(head * nblk, 400 + c) for c in range(4)])
    s = reevr_amd.ConvolverSet(nch)
    assert s.init(head, tail, irs, max_len=head), s.last_error_string
    assert s.subsets == 2 and s.stream(0), "the set's ordering stream must exist although child 0 has no stages"
    p = s.plan()
    assert p["live"] == 1 and p["tail_partitions"] >= 1 and p["subsets"] == 2
    dx = torch.from_numpy(x4).cuda()[torch.arange(nch, device="cuda") % 4].contiguous()
    out = torch.full_like(dx, float("nan"))
    for rep in range(3):                                  # producer and consumer on torch's stream, ordered through stream 0
        dx2 = dx * 1.0
        got = s.process_device_blocks(dx2, head, out, sync=False, order=True)
        total = got.abs().sum(dim=1)                      # consumer kernel on torch's stream, right behind the call
        s.clear()
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert s.last_error == 0, s.last_error_string
    s.close()
    assert np.all(got[:nch // 2] == 0) and np.all(total[:nch // 2].cpu().numpy() == 0)
    assert np.isfinite(got).all()
    for c in (nch // 2, nch // 2 + 1, nch - 1):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        assert rel_rms(got[c], o.process(x4[c % 4])) <= TOL, c


@pytest.mark.parametrize("widen", [False, True], ids=["tail8192", "tail16384"])
def test_row_looping_transforms_match_one_row_kernels(widen):
    """The row-looping 8192-bin transform kernels (k_fft8_fwd_loop / _inv_loop: many lock-step channels' tail jobs; the inverse only
    where it runs in float) against the one-row kernels on the same set, and against the oracle: 600 channels (more rows than
    resident workgroups, so workgroups really loop and prefetch), head 512 / tail 8192, short IRs with a tail stage. tail8192: the
    default of lock-step sets (forward float: loops; inverse double: one-row kernel either way); tail16384: the tail widened to
    twice the block (forced): one-row kernels in both modes, the knob must change nothing."""
    import torch
    nch, head, tail, nblk = 600, 512, 8192, 16 * (10 if widen else 7)
    irs = [synth.synth_ir(2 * tail + 3 * tail - 101 * (c % 7), 1, 40 + c % 11)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 300 + c % 9) for c in range(nch)])
    dx = torch.from_numpy(x).cuda()
    outs = {}
    for mode in (0, 1):
        reevr_amd.set_tuning("fft_loop", mode)
        try:
            s = reevr_amd.ConvolverSet(nch, tune={"tail_slack": 1} if widen else None)
            assert s.init(head, tail, irs, max_len=head), s.last_error_string
            assert s.tail_block == (2 * tail if widen else tail) and s.plan()["tail_f64"] == (0 if widen else 2)
            outs[mode] = s.process_device_blocks(dx, head).cpu().numpy()
            assert s.last_error == 0, s.last_error_string
            s.close()
        finally:
            reevr_amd.set_tuning("fft_loop", -1)
    assert np.isfinite(outs[1]).all()
    for c in range(nch):
        assert rel_rms(outs[1][c], outs[0][c]) <= 2e-6, c
    for c in (0, 299, 599):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        assert rel_rms(outs[1][c], o.process(x[c])) <= TOL, c


def test_double_inverse_as_two_half_transforms(golden):
    """k_fft8_inv_dif2 -- the 8192-bin inverse in double as two 4096-point sub-transforms in two workgroups, the tail inverse of
    lock-step sets -- against the one-row double kernel ("inv_dif" 0) and the known answers: (1) the bare
    transform on the reference's AudioFFT spectrum (tests/golden/audiofft.npz, n = 16384), both forms within the f64 tolerance
    of the golden round trip and within 1e-7 of each other; (2) a 601-channel set (an odd job count: the last chunk of 8 has
    unused slots), head 512 / tail 8192, both forms sample against sample and against the oracle."""
    import torch
    from reevr_amd import _lib
    L = _lib.lib()
    g = golden["audiofft"]
    n = 16384
    fp = lambda a: a.ctypes.data_as(_lib.F32P)
    wre = np.ascontiguousarray(g[f"n{n}/re"], np.float32)
    wim = np.ascontiguousarray(g[f"n{n}/im"], np.float32)
    wrt = g[f"n{n}/rt"].astype(np.float64)
    rts = {}
    for mode in (0, 1):
        reevr_amd.set_tuning("inv_dif", mode)
        try:
            rt = np.full(n, np.nan, np.float32)
            assert L.rvc_debug_irfft(0, n, 1, fp(rt), fp(wre), fp(wim)) == 1
            rts[mode] = rt.astype(np.float64)
        finally:
            reevr_amd.set_tuning("inv_dif", -1)
        assert np.sqrt(np.mean((rts[mode] - wrt) ** 2)) / np.sqrt(np.mean(wrt ** 2)) <= 3e-7, mode
    assert np.sqrt(np.mean((rts[1] - rts[0]) ** 2)) / np.sqrt(np.mean(wrt ** 2)) <= 1e-7
    nch, head, tail, nblk = 601, 512, 8192, 16 * 7
    irs = [synth.synth_ir(2 * tail + 3 * tail - 101 * (c % 7), 1, 40 + c % 11)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 300 + c % 9) for c in range(nch)])
    dx = torch.from_numpy(x).cuda()
    outs = {}
    for mode in (0, -1):
        s = reevr_amd.ConvolverSet(nch, tune={"inv_dif": mode})
        assert s.init(head, tail, irs, max_len=head), s.last_error_string
        assert s.tail_block == tail and s.plan()["tail_f64"] == 2
        outs[mode] = s.process_device_blocks(dx, head).cpu().numpy()
        assert s.last_error == 0, s.last_error_string
        s.close()
    assert np.isfinite(outs[-1]).all()
    for c in range(nch):
        assert rel_rms(outs[-1][c], outs[0][c]) <= 1e-6, c
    for c in (0, 300, 600):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        assert rel_rms(outs[-1][c], o.process(x[c])) <= TOL, c


@pytest.mark.parametrize("seed", [0, 3, 7, 11, 226])
def test_guard_bands_stay_intact_and_outputs_finite(seed):
    """Out-of-bounds net (rvc_debug_guard_check): every device allocation of the set between NaN-filled guard bands and
    NaN-poisoned itself. A block-synchronous fuzz run with forced two-level tiling, ragged / multi-block / long calls and a
    clear(): no guard byte may change (no out-of-bounds write), every output must be finite and match the oracle (a value
    read out of bounds or never written and USED would be a NaN: the 0ca535d bug class fails here, not by seed luck)."""
    rng = np.random.RandomState(9100 + seed)
    head = int(rng.choice([64, 128, 256, 512]))
    tail = int(rng.choice([2 * head, 4 * head, 16 * head]))
    nch = int(rng.randint(1, 4))
    parts = 3 if seed == 226 else int(rng.choice([1, 3, 9, 20]))
    base = 2 * tail + parts * tail - int(rng.randint(0, tail // 2))
    irs = [synth.synth_ir(max(1, base - c * int(rng.randint(0, tail))), 1, 800 + 3 * seed + c)[0] for c in range(nch)]
    total = int(min(max(40 * tail, 30 * 8 * head), 200000))
    total -= total % head
    sched, done = [], 0
    while done < total:
        r = rng.randint(0, 30)
        n = (int(rng.randint(1, head)) if r == 0 else (head - done % head) if (r == 1 and done % head) else
             int(rng.randint(2, 6)) * head if r == 2 else int(rng.randint(5, 9)) * tail if r == 3 else
             (head if done % head == 0 else head - done % head))
        n = max(1, min(n, total - done))
        sched.append(n)
        done += n
    x = np.stack([synth.synth_input(total, 13 * seed + c) for c in range(nch)])
    reevr_amd.set_tuning("guard", 1)
    try:
        s = reevr_amd.ConvolverSet(nch, bg_stream=bool(seed & 1), time_tiling="force2")
        assert s.init(head, tail, irs, max_len=max(sched)), s.last_error_string
    finally:
        reevr_amd.set_tuning("guard", 0)
    assert s.guard_check() == 0
    got = np.empty_like(x)
    pos = 0
    for n in sched:
        got[:, pos:pos + n] = s.process(x[:, pos:pos + n])
        pos += n
    assert s.last_error == 0, s.last_error_string
    assert s.guard_check() == 0, "a kernel wrote outside its allocation"
    assert np.isfinite(got).all(), "a kernel used a value it read out of bounds / that nobody wrote"
    for c in range(nch):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        assert rel_rms(got[c], o.process(x[c])) <= TOL, c
    s.close()


def test_guard_mode_lockstep_many_channels():
    """The same net on the lock-step regime's kernels: 96 channels, head 512 / tail 8192 (2-wave per-block kernel, sweeps,
    patches, tail transforms), default tiling by size + forced; guards intact, outputs finite, three channels vs the oracle."""
    import torch
    nch, head, tail, nblk = 96, 512, 8192, 16 * 6
    irs = [synth.synth_ir(2 * tail + 19 * tail - 977 * (c % 5), 1, 600 + c)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 20 + c % 7) for c in range(nch)])
    dx = torch.from_numpy(x).cuda()
    for tiling in (True, "force2"):
        reevr_amd.set_tuning("guard", 1)
        try:
            s = reevr_amd.ConvolverSet(nch, time_tiling=tiling)
            assert s.init(head, tail, irs, max_len=head), s.last_error_string
        finally:
            reevr_amd.set_tuning("guard", 0)
        got = s.process_device_blocks(dx, head).cpu().numpy()
        assert s.last_error == 0, s.last_error_string
        assert s.guard_check() == 0
        assert np.isfinite(got).all()
        for c in (0, 47, 95):
            o = O.TwoStageFFTConvolver("orc")
            assert o.init(head, tail, irs[c])
            assert rel_rms(got[c], o.process(x[c])) <= TOL, (tiling, c)
        s.close()


@pytest.mark.parametrize("head,block", [(128, 128), (256, 256), (128, 100)])
def test_two_wave_block_kernel_small_heads(head, block):
    """Head blocks of 128 / 256: 4 / 2 channels share one audio wave and its patch wave (k_fused_block2w<7>, <8>); channel
    counts that leave a workgroup partly empty; tiled by force, against the oracle."""
    import torch
    tail = 8 * head
    for nch in (1, 3, 5):
        irs = [synth.synth_ir(2 * tail + 6 * tail - 13 * c, 1, 70 + c)[0] for c in range(nch)]
        nblk = 200
        x = np.stack([synth.synth_input(block * nblk, 80 + c) for c in range(nch)])
        s = reevr_amd.ConvolverSet(nch, time_tiling="force")
        assert s.init(block, tail, irs, max_len=block), s.last_error_string
        got = s.process_device_blocks(torch.from_numpy(x).cuda(), block).cpu().numpy()
        assert s.last_error == 0, s.last_error_string
        s.close()
        for c in range(nch):
            o = O.TwoStageFFTConvolver("orc")
            assert o.init(block, tail, irs[c])
            assert rel_rms(got[c], o.process(x[c])) <= TOL, (nch, c)


@pytest.mark.parametrize("fft_many", [0, 1])
def test_many_channels_large_head_block_general_path(fft_many):
    """Many lock-step channels with a LARGE head block (BASELINE config 5's geometry: head 4096 / tail 8192): per-block
    calls are served by transform / delay-line / inverse launches instead of the one-workgroup-per-channel latency kernel
    (rvc_state.cpp block_general; the head transform reads the block from the caller's buffer and appends it to the ring,
    the tail job runs behind it); tail stage time-tiled; block calls, then a ragged pair, against the oracle. fft_many: the
    many-rows form of the 4096-bin transforms (twiddles per pass: launches of >= 2048 rows by default) forced on / off."""
    import torch
    nch, head, tail, nblk = 256, 4096, 8192, 40
    irs = [synth.synth_ir(2 * tail + 17 * tail - 333 * (c % 5), 1, 30 + c % 13)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 60 + c % 7) for c in range(nch)])
    dx = torch.from_numpy(x).cuda()
    with reevr_amd.tuning(fft_many=fft_many):
        s = reevr_amd.ConvolverSet(nch)
        assert s.init(head, tail, irs, max_len=head), s.last_error_string
        assert s.tile_rows(1) == 8
        a = s.process_device_blocks(dx[:, :head * 38].contiguous(), head)
        b = s.process_device(dx[:, head * 38:head * 38 + 1000].contiguous())
        c = s.process_device(dx[:, head * 38 + 1000:head * 39].contiguous())
        d = s.process_device_blocks(dx[:, head * 39:].contiguous(), head)
        got = torch.cat([a, b, c, d], dim=1).cpu().numpy()
        assert s.last_error == 0, s.last_error_string
        s.close()
    for ch in (0, 100, 255):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[ch])
        assert rel_rms(got[ch], o.process(x[ch])) <= TOL, ch


def _fdl_case(rng, nch, B, P, M, rows):
    """split-complex operands the way the reference holds them (B + 1 bins, im[0] = im[B] = 0: SplitComplex of a real
    transform) and their packed interleaved form (bin 0 = (DC, Nyquist)) for the GPU"""
    def split(shape):
        re = rng.uniform(-1, 1, shape + (B + 1,)).astype(np.float32)
        im = rng.uniform(-1, 1, shape + (B + 1,)).astype(np.float32)
        im[..., 0] = 0.0
        im[..., B] = 0.0
        return re, im

    def pack(re, im):
        out = np.empty(re.shape[:-1] + (B, 2), np.float32)
        out[..., 0] = re[..., :B]
        out[..., 1] = im[..., :B]
        out[..., 0, 1] = re[..., B]
        return np.ascontiguousarray(out)
    return split, pack


@pytest.mark.parametrize("kind,nch,B,P,M,delay,k0,variant", [
    (0, 3, 512, 6, 1, 0, 40, "patch"),            # k_fdl_patch: a sweep row + a few recent partitions
    (0, 2, 8192, 10, 1, 2, 40, "patch"),
    (0, 600, 512, 4, 1, 0, 9, "row_many"),         # many channels, no base row: the streaming single-row form
    (0, 2, 512, 33, 1, 0, 100, "row"),             # k_fir_row (the latency-oriented row kernel), whole delay line
    (0, 2, 512, 33, 1, 0, 5, "row_early"),         # rows before block 0 read as zero
    (0, 2, 256, 7, 5, 0, 20, "fir"),               # k_fir<TK>: a few rows per launch
    (0, 2, 1024, 19, 12, 2, 30, "fir"),
    (0, 2, 512, 21, 40, 0, 64, "fir_lds"),         # k_fir_lds: long calls, LDS-tiled
    (1, 2, 8192, 20, 8, 2, 64, "split"),           # K = 8 sweep, partition-split form
    (1, 3, 512, 30, 8, 0, 64, "own"),              # K = 8 sweep, own-tile form
    (1, 3, 512, 17, 8, 0, 64, "second"),           # second-level sweep: window [x_from, x_hi] + first-level rows
    (1, 2, 8192, 57, 16, 2, 128, "own"),           # first-level sweeps of long delay lines
    (1, 3, 256, 64, 16, 0, 128, "own"),
    (1, 2, 512, 94, 32, 0, 128, "own"),
    (1, 2, 8192, 11, 16, 2, 3, "own_early"),       # the clock has just started: most rows do not exist yet
    # round 4: the LDS-fed first-level sweeps (accumulators split over waves that share every fetched row through LDS-DMA rings)
    (1, 2, 512, 94, 32, 0, 128, "lds32"),          # 2 x 16 blocks (the default form of 32-block tiles), config 1's line
    (1, 3, 8192, 37, 32, 2, 200, "lds32"),         # tail-stage rows, a partial last body (37 = 2 x 16 + 5), three channels
    (1, 2, 128, 5, 32, 0, 64, "lds32"),            # one 128-bin piece per row, fewer partitions than a ring holds
    (1, 2, 256, 1, 32, 2, 64, "lds32"),            # a single partition
    (1, 2, 512, 40, 32, 2, 7, "lds32_early"),      # the clock has just started: rows before block 0 read as zero
    (1, 2, 512, 50, 32, 0, 96, "lds32_allrows"),   # every window row counts (x_hi beyond the tile): the windows' own loads
    (1, 2, 1024, 45, 32, 2, 96, "lds32_4x8"),      # 4 x 8 blocks (512-thread workgroups)
    (1, 2, 512, 50, 32, 0, 96, "lds32_4x8_allrows"),
    (1, 2, 512, 41, 32, 2, 96, "lds32_deep"),      # rings three chunks ahead (48 KiB)
    (1, 2, 8192, 57, 16, 2, 128, "lds16"),         # 16-block tiles as 2 x 8
    (1, 3, 256, 23, 16, 0, 40, "lds16_allrows"),
    (1, 2, 512, 94, 32, 0, 128, "one_wave32"),     # the one-wave 32-block form (sweep_lds = 0)
    # round 4: the delay-1 tail stage of many-channel sets (the newest row a sweep may use is k0 - 1), 16384-bin rows
    (0, 2, 16384, 6, 1, 1, 40, "patch_d1"),
    (1, 2, 8192, 20, 8, 1, 64, "split_d1"),
    (1, 3, 512, 17, 8, 1, 64, "second_d1"),
    (1, 2, 16384, 29, 16, 1, 64, "own_d1"),
    (1, 2, 16384, 40, 32, 1, 100, "lds32_d1"),
    (1, 2, 1024, 11, 16, 1, 2, "own_early_d1"),
    # round 5: the three-product complex multiply-accumulate of the LDS-fed 32-block sweeps (S1 / S2 / S3 accumulators, combined at
    # the end of the walk; the packed DC / Nyquist bin through per-lane operands) -- forced on (_m3) and off (_m4) whatever ships
    (1, 2, 512, 94, 32, 0, 128, "lds32_m3"),
    (1, 3, 8192, 37, 32, 2, 200, "lds32_m3"),
    (1, 2, 128, 5, 32, 0, 64, "lds32_m3"),
    (1, 2, 256, 1, 32, 2, 64, "lds32_m3"),
    (1, 2, 512, 40, 32, 2, 7, "lds32_early_m3"),
    (1, 2, 512, 50, 32, 0, 96, "lds32_allrows_m3"),
    (1, 2, 1024, 45, 32, 2, 96, "lds32_4x8_m3"),
    (1, 2, 16384, 40, 32, 1, 100, "lds32_m3_d1"),
    (1, 2, 512, 94, 32, 0, 128, "lds32_m4"),
    (1, 3, 8192, 37, 32, 2, 200, "lds32_m4"),
    (1, 2, 16384, 40, 32, 1, 100, "lds32_m4_d1"),
])
def test_delay_line_kernels_in_isolation(kind, nch, B, P, M, delay, k0, variant):
    """The complex multiply-accumulate kernels ALONE (SURVEY a-12): one launch of the general delay-line launcher / a sweep
    on given rows through rvc_debug_fdl, against the reference's ComplexMultiplyAccumulate (Utilities.cpp:62-111, restated
    in oracle/rvc_oracle.c and pinned bit for bit by tests/test_oracle.py::test_cmac_vs_golden) applied partition by
    partition the way FFTConvolver.cpp:176-187 applies it. Tolerance: the GPU accumulates with fused multiply-adds and, in
    the partition-split sweep, in four partial sums."""
    import ctypes as C
    from reevr_amd import _lib as L
    rng = np.random.RandomState(1000 * kind + B + P + M)
    rows = 1
    while rows < P + M + 4:
        rows *= 2
    split, pack = _fdl_case(rng, nch, B, P, M, rows)
    Hre, Him = split((nch, P))
    Xre, Xim = split((nch, rows))
    d1 = variant.endswith("_d1")
    variant = variant[:-3] if d1 else variant
    mac3 = {"_m3": 1, "_m4": 0}.get(variant[-3:], -1)
    variant = variant[:-3] if mac3 >= 0 else variant
    second = variant == "second"
    has_add = variant in ("patch", "second")
    Are, Aim = split((nch, M if kind == 1 else 1)) if has_add else (None, None)
    x_hi = k0 + M - 1 - delay if kind == 0 else (k0 + M if variant.endswith("allrows") else ((k0 - 1 if d1 else k0 - 2) if variant != "second" else k0 + 3))
    x_from = (k0 - 9) if second else 0
    want_re = np.zeros((nch, M, B + 1), np.float32)
    want_im = np.zeros((nch, M, B + 1), np.float32)
    for c in range(nch):
        for m in range(M):
            re = Are[c, m if kind == 1 else 0].copy() if has_add else np.zeros(B + 1, np.float32)
            im = Aim[c, m if kind == 1 else 0].copy() if has_add else np.zeros(B + 1, np.float32)
            for i in range(P):
                row = k0 + m - delay - i
                if row < 0 or (kind == 1 and not (x_from <= row <= x_hi)):
                    continue
                O.cmac(re, im, Hre[c, i], Him[c, i], Xre[c, row & (rows - 1)], Xim[c, row & (rows - 1)])
            want_re[c, m], want_im[c, m] = re, im
    want = pack(want_re, want_im)
    H, X = pack(Hre, Him), pack(Xre, Xim)
    A = pack(Are, Aim) if has_add else None
    got = np.empty((nch, M, B, 2), np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None
    tune = {"split": 1, "own": 0, "own_early": 0, "second": 0}.get(variant)
    lds = 2 if variant.startswith("lds32_4x8") else (3 if variant.startswith("lds16") else (0 if variant == "one_wave32" else (1 if variant.endswith("deep") else -1)))
    with reevr_amd.tuning(sweep_lds=lds, mac3=mac3, **({"sweep_split": tune} if tune is not None else {})):
        ok = L.lib().rvc_debug_fdl(0, kind, nch, B, P, M, delay, k0, rows, fp(H), fp(X), fp(A), fp(got), x_hi, x_from)
    assert ok == 1
    if kind == 1:                                   # output row j sits in slot (k0 + j) & (M - 1)
        got = np.stack([got[:, (k0 + j) & (M - 1)] for j in range(M)], axis=1)
    assert np.isfinite(got).all()
    d = got.astype(np.float64) - want
    scale = max(float(np.abs(want).max()), 1e-9)
    assert np.abs(d).max() <= 1e-5 * scale, (variant, float(np.abs(d).max()), scale)
    assert np.sqrt(np.mean(d ** 2)) <= 2e-6 * np.sqrt(np.mean(want.astype(np.float64) ** 2))



@pytest.mark.parametrize("block,nblocks,ir_len,quad", [(512, 40, 30000, True), (480, 30, 20000, False), (64, 50, 900, True)])
def test_reference_stereo_convolver_glue_on_the_drop_in(tmp_path, block, nblocks, ir_len, quad):
    """INTEGRATION.md's "compiles unchanged" claim, run: the REFERENCE's own src/dsp/StereoConvolver.cpp (compiled where it lies
    against include/reevr_amd/Convolver.h by __graft_entry__.build() -> oracle/_ref/ref_glue; tests/test_abi.py has the recipe)
    drives four drop-in Convolvers through prepare -> loadImpulse -> process per block -> clear -> process, and its public
    buffers must be what four oracle TwoStageFFTConvolvers give (StereoConvolver.cpp:22-62; LR <- L, RL <- R; with a stereo
    impulse the LR / RL buffers stay untouched)."""
    import subprocess
    from tests import test_abi
    exe = test_abi.GLUE_EXE
    if not os.path.exists(exe):
        if not os.path.exists(test_abi.REF_SC):
            pytest.skip("oracle/_ref/ref_glue was not built (needs /root/reference at build time)")
        exe = test_abi.build_reference_glue(str(tmp_path))
    pin, pout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    irs, x = test_abi.glue_operands(pin, block, nblocks, ir_len, quad)
    r = subprocess.run([exe, pin, pout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    got = np.fromfile(pout, np.float32).reshape(nblocks, 4, block).transpose(1, 0, 2).reshape(4, nblocks * block)
    head = 1
    while head < block:
        head *= 2
    cut = (nblocks // 2 + 1) * block                       # clear() behind block nblocks / 2
    for c, src in enumerate([0, 1, 0, 1]):                 # LL <- L, RR <- R, LR <- L, RL <- R
        if c >= 2 and not quad:
            assert np.all(got[c] == 0)
            continue
        o = O.TwoStageFFTConvolver()
        assert o.init(head, max(8192, 2 * head), irs[c])
        want = np.concatenate([_stream(o, x[src, :cut], block), (o.clear(), _stream(o, x[src, cut:], block))[1]])
        if cut % head:                                     # (a clear() inside a head block: the reference's stale-accumulator quirk,
            continue                                       #  SURVEY a-11 -- not part of the parity contract)
        assert rel_rms(got[c], want) <= 1e-5, (c, rel_rms(got[c], want))


def _stream(o, x, block):
    return np.concatenate([o.process(x[a:a + block]) for a in range(0, len(x), block)]) if len(x) else np.zeros(0, np.float32)


@pytest.mark.parametrize("nch,zc", [(2, -1), (6, 0), (600, -1), (600, 1)])
def test_host_staging_rows_in_place(nch, zc):
    """rvc_set_host_buffers: a host that writes its block into the set's own pinned staging rows and passes THOSE pointers to
    rvc_set_process gets the same samples as through its own buffers -- no staging copy (600 channels: the many-thread staging
    path for the ordinary call, DMA or in-kernel PCIe access by knob) -- per-block calls, a ragged call, and the ordinary
    host-pointer call on the same handle in between; two channels against the oracle."""
    head, tail, nblk = 128, 512, 40
    irs = [synth.synth_ir(2 * tail + 3 * tail - 7 * (c % 13), 1, 40 + c % 9)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 5 + c % 7) for c in range(nch)])
    s = reevr_amd.ConvolverSet(nch, tune=dict(host_zero_copy=zc))
    assert s.init(head, tail, irs, max_len=head), s.last_error_string
    ins, outs = s.host_buffers()
    got = np.empty_like(x)
    pos = 0
    sched = [head] * 10 + [37, head - 37] + [head] * 8 + [-head] * 4 + [head] * (nblk - 23)      # (negative: through own buffers)
    for n in sched:
        if n < 0:
            n = -n
            got[:, pos:pos + n] = s.process(x[:, pos:pos + n])
        else:
            for c in range(nch):
                ins[c][:n] = x[c, pos:pos + n]
            s.process_in_place(n)
            for c in range(nch):
                got[c, pos:pos + n] = outs[c][:n]
        pos += n
    assert pos == head * nblk and s.last_error == 0, s.last_error_string
    s.clear()
    ref = np.concatenate([s.process(x[:, b * head:(b + 1) * head]) for b in range(nblk)], axis=1)
    s.close()
    assert np.array_equal(got[:, :10 * head], ref[:, :10 * head])           # same calls, other buffers: the same bits
    for c in range(0, nch, max(1, nch // 16)):                              # (behind the ragged calls: the same samples)
        assert rel_rms(got[c], ref[c]) <= 1e-6, c
    for c in (0, nch - 1):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        assert rel_rms(got[c], o.process(x[c])) <= TOL, c


def test_per_block_kernel_lane_local_exchange_is_bit_identical():
    """The per-block kernel of head 512 with the second exchange of its two transforms done LANE-LOCALLY (v_permlane32_swap /
    v_permlane16_swap / DPP row_ror:8: knob block_lanex, rvc_kernels.hip lanex_transpose) instead of through LDS: a pure
    permutation, so the same bits -- 64 lock-step channels (time-tiled zero-latency stage: the two-wave kernel), per-block calls
    incl. ragged ones through the host entry; one channel against the oracle."""
    import torch
    nch, head, tail, ir_len, nblk = 64, 512, 8192, 70000, 120
    irs = [synth.synth_ir(ir_len, 1, 500 + c)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 40 + c) for c in range(nch)])
    dx = torch.from_numpy(x).cuda()
    outs = []
    for lx in (0, 1):
        s = reevr_amd.ConvolverSet(nch, tune=dict(block_lanex=lx))
        assert s.init(head, tail, irs, max_len=head)
        assert s.plan()["head_patch_in_launch"] == 1
        y = s.process_device_blocks(dx[:, :head * 100].contiguous(), head).cpu().numpy()
        z = np.concatenate([s.process(x[:, a:b]) for a, b in ((head * 100, head * 100 + 37), (head * 100 + 37, head * 101),
                                                                (head * 101, head * 102))], axis=1)
        assert s.last_error == 0
        s.close()
        outs.append(np.concatenate([y, z], axis=1))
    assert np.array_equal(outs[0], outs[1])
    o = O.TwoStageFFTConvolver("orc")
    assert o.init(head, tail, irs[5])
    assert rel_rms(outs[1][5], o.process(x[5, :head * 102])) <= TOL


def test_float_inverse_16384_as_two_half_transforms(golden):
    """k_fft8_inv_dif2<13, float> and k_fft8_fwd_dif2<13> (knobs inv_dif14 / fwd_dif14; round 6, review item 3b) -- the 16384-bin
    float inverse and forward as two 8192-point sub-transforms in two workgroups of 1024 threads (a half-size plan WITH a final
    radix-2 pass) -- against the whole-CU one-row kernels and the known answers: (1) the bare transform on the reference's AudioFFT spectrum (n = 32768), both forms within the
    float tolerance of the golden round trip and within 1e-6 of each other; (2) a 301-channel set whose tail runs at block 16384
    (head 256 / tail 8192, widened: config 3's structure; an odd job count), both forms against each other and the oracle."""
    import torch
    from reevr_amd import _lib
    L = _lib.lib()
    g = golden["audiofft"]
    n = 32768
    fp = lambda a: a.ctypes.data_as(_lib.F32P)
    wre = np.ascontiguousarray(g[f"n{n}/re"], np.float32)
    wim = np.ascontiguousarray(g[f"n{n}/im"], np.float32)
    wrt = g[f"n{n}/rt"].astype(np.float64)
    rts = {}
    for mode in (0, 1):
        reevr_amd.set_tuning("inv_dif14", mode)
        try:
            rt = np.full(n, np.nan, np.float32)
            assert L.rvc_debug_irfft(0, n, 0, fp(rt), fp(wre), fp(wim)) == 1
            rts[mode] = rt.astype(np.float64)
        finally:
            reevr_amd.set_tuning("inv_dif14", -1)
        assert np.sqrt(np.mean((rts[mode] - wrt) ** 2)) / np.sqrt(np.mean(wrt ** 2)) <= 2e-6, mode
    assert np.sqrt(np.mean((rts[1] - rts[0]) ** 2)) / np.sqrt(np.mean(wrt ** 2)) <= 1e-6     # (two float forms: measured 2.6e-7)
    # ... and the FORWARD transform the same way (k_fft8_fwd_dif2<13>, knob fwd_dif14): the reference's spectrum of seeded noise
    x = synth.white_noise(n, 0xF00D + n)
    want = np.concatenate([g[f"n{n}/re"], g[f"n{n}/im"]]).astype(np.float64)
    specs = {}
    for mode in (0, 1):
        reevr_amd.set_tuning("fwd_dif14", mode)
        try:
            re, im = np.full(n // 2 + 1, np.nan, np.float32), np.full(n // 2 + 1, np.nan, np.float32)
            assert L.rvc_debug_rfft(0, n, 0, fp(x), fp(re), fp(im)) == 1
            specs[mode] = np.concatenate([re, im]).astype(np.float64)
        finally:
            reevr_amd.set_tuning("fwd_dif14", -1)
        assert np.sqrt(np.mean((specs[mode] - want) ** 2)) / np.sqrt(np.mean(want ** 2)) <= 2e-6, mode
        assert specs[mode][n // 2 + 1] == 0.0 and specs[mode][-1] == 0.0                        # im[0] = im[n/2] = 0 (AudioFFT.cpp:134-135)
    assert np.sqrt(np.mean((specs[1] - specs[0]) ** 2)) / np.sqrt(np.mean(want ** 2)) <= 1e-6
    nch, head, tail, nblk = 301, 256, 8192, 64 * 6
    irs = [synth.synth_ir(2 * tail + 5 * tail - 101 * (c % 7), 1, 40 + c % 11)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 300 + c % 9) for c in range(nch)])
    dx = torch.from_numpy(x).cuda()
    outs = {}
    for mode in (0, 1):
        s = reevr_amd.ConvolverSet(nch, tune={"inv_dif14": mode, "fwd_dif14": mode, "tail_slack": 1})
        assert s.init(head, tail, irs, max_len=head), s.last_error_string
        assert s.tail_block == 2 * tail and s.plan()["tail_f64"] == 0
        outs[mode] = s.process_device_blocks(dx, head).cpu().numpy()
        assert s.last_error == 0, s.last_error_string
        s.close()
    assert np.isfinite(outs[1]).all()
    for c in range(nch):
        assert rel_rms(outs[1][c], outs[0][c]) <= 1e-6, c
    for c in (0, 150, 300):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        assert rel_rms(outs[1][c], o.process(x[c])) <= TOL, c
