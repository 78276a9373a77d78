"""CPU tests (-m "not gpu"): pin the oracle (oracle/rvc_oracle.c) against
  * the committed golden fixtures generated from the untouched reference (tests/golden),
  * the reference's own 58 known-answer cases and pass rule (Test.cpp:129-145, :256-329),
  * oracle/_ref itself when it has been built in this container.
"""
import os
import numpy as np
import pytest

from oracle import oracle_py as O
from tests import cases
from tests.conftest import fixture_of

# oracle vs reference: both are exact-to-double DFTs rounded to float at the same places
ORACLE_TOL = 1e-6


def orc_factory(kind):
    return O.FFTConvolver("orc") if kind == "fftconv" else O.TwoStageFFTConvolver("orc")


def ref_factory(kind):
    return O.FFTConvolver("ref") if kind == "fftconv" else O.TwoStageFFTConvolver("ref")


KATS = [("fftconv", t) for t in cases.KAT_FFTCONV] + [("twostage", t) for t in cases.KAT_TWOSTAGE]


@pytest.mark.parametrize("kind,tup", KATS, ids=[cases.kat_name(k, t) for k, t in KATS])
def test_kat_oracle_vs_golden_and_direct(golden, kind, tup):
    out = cases.run_kat(orc_factory, kind, tup)
    cases.compare_to_fixture(out, fixture_of(golden["kat"], cases.kat_name(kind, tup)), ORACLE_TOL)
    from reevr_amd import synth
    exact = O.direct_convolve(synth.ramp(tup[0]), synth.ramp(tup[1]))
    assert cases.kat_tolerance_ok(out, exact, tup[1])


@pytest.mark.parametrize("name", list(cases.SYNTH_CASES))
def test_synth_oracle_vs_golden(golden, name):
    out = cases.run_synth_case(orc_factory, cases.SYNTH_CASES[name])
    for c in range(out.shape[0]):
        cases.compare_to_fixture(out[c], fixture_of(golden["synth"], f"{name}/ch{c}"), ORACLE_TOL)


@pytest.mark.parametrize("n", [2, 4, 8, 16, 64, 1024, 16384, 32768])
def test_audiofft_oracle_vs_golden(golden, n):
    from reevr_amd import synth
    x = synth.white_noise(n, 0xF00D + n)
    re, im = O.rfft(x, "orc")
    g = golden["audiofft"]
    scale = np.abs(g[f"n{n}/re"]).max() + 1e-30
    assert np.abs(re - g[f"n{n}/re"]).max() <= 2e-7 * scale
    assert np.abs(im - g[f"n{n}/im"]).max() <= 2e-7 * scale
    # numpy.fft.rfft convention
    want = np.fft.rfft(x.astype(np.float64))
    assert np.abs(re - want.real).max() <= 2e-7 * scale
    assert np.abs(im - want.imag).max() <= 2e-7 * scale
    assert im[0] == 0.0 and im[-1] == 0.0
    rt = O.irfft(re, im, "orc")
    assert np.abs(rt - g[f"n{n}/rt"]).max() <= 4e-7
    assert np.abs(rt - x).max() <= 1e-6


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (no reference sources here)")
@pytest.mark.parametrize("name", ["small_three_stage", "ragged_calls_b512", "clear_block_aligned",
                                  "cfg4_inst3_10s_b512"])
def test_oracle_vs_live_reference(name):
    case = cases.SYNTH_CASES[name]
    a = cases.run_synth_case(orc_factory, case)
    b = cases.run_synth_case(ref_factory, case)
    d = a.astype(np.float64) - b.astype(np.float64)
    assert np.sqrt(np.mean(d ** 2)) <= 1e-7
    assert np.abs(d).max() <= 1e-6


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_clear_mid_block_quirk_matches_reference():
    """FFTConvolver::clear keeps _inputBufferFill and _preMultiplied (FFTConvolver.cpp:80-90);
    the oracle restates that quirk, so a mid-block clear() must match the reference too."""
    from reevr_amd import synth
    ir = synth.synth_ir(1500, 1, 3)[0]
    x = synth.synth_input(64 * 50, 0)
    outs = []
    for which in ("orc", "ref"):
        c = O.TwoStageFFTConvolver(which)
        assert c.init(64, 256, ir)
        o = np.empty_like(x)
        pos = 0
        for i in range(len(x) // 40):
            if i == 33:
                c.clear()
            o[pos:pos + 40] = c.process(x[pos:pos + 40])
            pos += 40
        outs.append(o[:pos])
    assert np.abs(outs[0] - outs[1]).max() <= 1e-6


def test_init_error_and_empty_semantics():
    """init -> False iff a block size is 0; empty / all-zero IR -> True and zeros out
    (FFTConvolver.cpp:97-111, TwoStageFFTConvolver.cpp:94-115); process before init = zeros."""
    ir = np.ones(10, np.float32)
    c = O.FFTConvolver("orc")
    assert np.all(c.process(np.ones(7, np.float32)) == 0)
    assert c.init(0, ir) is False
    assert c.init(4, np.zeros(10, np.float32)) is True
    assert np.all(c.process(np.ones(7, np.float32)) == 0)
    t = O.TwoStageFFTConvolver("orc")
    assert t.init(0, 8, ir) is False and t.init(8, 0, ir) is False
    assert t.init(4, 8, np.zeros(0, np.float32)) is True
    assert np.all(t.process(np.ones(9, np.float32)) == 0)
    # head > tail is swapped (TwoStageFFTConvolver.cpp:100-104)
    a = O.TwoStageFFTConvolver("orc"); a.init(16, 4, ir)
    b = O.TwoStageFFTConvolver("orc"); b.init(4, 16, ir)
    x = np.arange(40, dtype=np.float32)
    assert np.array_equal(a.process(x), b.process(x))


def _cmac_inputs(n):
    from reevr_amd import synth
    return [synth.white_noise(n, 0xC0AC + 16 * n + j) for j in range(6)]     # (oracle/gen_golden.py cmac_inputs)


@pytest.mark.parametrize("n", [1, 3, 4, 7, 64, 513, 1027])
def test_cmac_vs_golden(n):
    """ComplexMultiplyAccumulate (Utilities.cpp:62-111) alone: the oracle's restatement against the reference's own
    result (tests/golden/cmac.npz, generated by oracle/gen_golden.py from oracle/_ref), bit for bit -- the SSE order
    of the first 4 * (len / 4) values and the scalar tail included."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cmac.npz"))
    a = _cmac_inputs(n)
    re, im = a[0].copy(), a[1].copy()
    O.cmac(re, im, *a[2:], which="orc")
    assert np.array_equal(re, g[f"n{n}/re"]) and np.array_equal(im, g[f"n{n}/im"])


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_cmac_vs_live_reference():
    rng = np.random.RandomState(5)
    for n in (2, 5, 8, 129, 4097):
        a = [rng.randn(n).astype(np.float32) for _ in range(6)]
        r1, i1, r2, i2 = a[0].copy(), a[1].copy(), a[0].copy(), a[1].copy()
        O.cmac(r1, i1, *a[2:], which="orc")
        O.cmac(r2, i2, *a[2:], which="ref")
        assert np.array_equal(r1, r2) and np.array_equal(i1, i2), n

