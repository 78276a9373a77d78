"""CPU tests of the impulse-preparation restatement (oracle/impulse_oracle.c)
against the committed fixture (made with the reference's AudioFFT in the STFT stage), properties
that hold for any correct implementation, and the host-side pieces of the product
(rvc_impulse_decay_lut, error behaviour without a GPU)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_py as O  # noqa: E402
from tests import impulse_cases as IC  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "impulse.npz"))


def oracle_case(case, fft="orc"):
    n, nc, seed, kw, mag, rate = IC.params_of(case)
    lut = None if mag is None else O.impulse_decay_lut(mag, kw["srate"], rate)
    return O.impulse_recalc(IC.raw_channels(n, nc, seed), decay_lut=lut, fft=fft, **kw), lut


@pytest.mark.parametrize("name", sorted(IC.CASES))
def test_oracle_matches_fixture(name):
    r, lut = oracle_case(IC.CASES[name])
    nc = IC.CASES[name][1]
    meta = GOLD[f"{name}/meta"]
    assert np.float32(r["peak"]) == np.float32(meta[0])
    assert (r["trim_left_samples"], r["trim_right_samples"]) == (int(meta[1]), int(meta[2]))
    if lut is not None:
        assert np.array_equal(lut, GOLD[f"{name}/lut"])
    for c in range(nc):
        want = GOLD[f"{name}/ch{c}"]
        got = r["buffers"][c]
        assert got.shape == want.shape
        # the fixture ran on the reference's Ooura transform, this run on the oracle's radix-2 one:
        # both double inside, so at most the last float bit of a sample can differ
        if want.size:
            assert np.max(np.abs(got.astype(np.float64) - want)) <= 2e-7 * max(np.max(np.abs(want)), 1e-30) + 1e-12


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_fixture_reproducible_with_reference_fft():
    for name in ("decay2", "decay4_all"):
        r, _ = oracle_case(IC.CASES[name], fft="ref")
        for c, b in enumerate(r["buffers"]):
            assert np.array_equal(b, GOLD[f"{name}/ch{c}"])


def test_lengths_and_empty():
    r, _ = oracle_case(IC.CASES["all_trimmed"])
    assert all(b.size == 0 for b in r["buffers"]) and r["trim_left_samples"] == 0
    r, _ = oracle_case(IC.CASES["trim_rev4"])
    assert len(r["buffers"]) == 4 and r["buffers"][0].size == 9000 - 900 - 2250
    assert r["trim_left_samples"] == 900 and r["trim_right_samples"] == 2250


def test_auto_gain_normalises_energy():
    raw = [4.0 * x for x in IC.raw_channels(6000, 2, 30)]
    r = O.impulse_recalc(raw, stage="a")
    e = sum(float(np.sum(b.astype(np.float64) ** 2)) for b in r["buffers"])
    assert abs(e - 1.0) < 1e-6                      # Impulse.cpp:691-708: 1/sqrt(energy), only when > 1
    quiet = [0.001 * x for x in raw]
    r = O.impulse_recalc(quiet, stage="a")
    assert np.array_equal(r["buffers"][0], quiet[0])   # never boosts
    assert r["peak"] == pytest.approx(float(max(np.max(np.abs(q)) for q in quiet)))


def test_reverse_and_trim_are_index_maps():
    raw = IC.raw_channels(5000, 2, 31)
    a = O.impulse_recalc(raw, stage="a")["buffers"]
    b = O.impulse_recalc(raw, stage="a", reverse=True, trim_left=0.2, trim_right=0.1)["buffers"]
    for c in range(2):
        assert np.array_equal(b[c], a[c][::-1][1000:4500])


def test_flat_decay_is_identity():
    """mag = 1 -> 0 dB -> decay factor exactly 1 for every bin: the STFT stage must give the input back
    (windowed overlap-add divided by the window sum)."""
    lut = O.impulse_decay_lut(np.ones(IC.LUT_SIZE, np.float32), 48000.0, 1.0)
    assert np.all(lut == 1.0)
    raw = IC.raw_channels(20000, 2, 32)
    a = O.impulse_recalc(raw, stage="a")["buffers"]
    b = O.impulse_recalc(raw, decay_lut=lut)["buffers"]
    for c in range(2):
        # the Blackman window is ~0.09 (2 pi i / 4096)^2 near i = 0, where only one frame contributes:
        # the float rounding of the spectrum is divided by it, so the first few hundred samples of
        # the reference's own output carry amplified rounding noise (up to ~1e-3 relative at i < 10)
        # (window[0] is ~3e-9: sample 0 is essentially that noise)
        pk = np.max(np.abs(a[c]))
        d = np.abs(a[c] - b[c]) / pk
        assert d[512:].max() < 1e-6 and d[64:512].max() < 5e-5 and d[8:64].max() < 5e-3


def test_decay_lut_formula():
    """Impulse.cpp:561-590 re-derived in numpy float32/float64."""
    mag = IC.MAGS["tilt"]
    srate, rate = 44100.0, 1.5
    lut = O.impulse_decay_lut(mag, srate, rate)
    lnD = np.log(np.power(1.0 - np.float64(np.float32(0.9)), (4096 / srate) * np.float32(rate)))
    lnG = np.log(np.power(3.0, (4096 / srate) * np.float32(rate)))
    dB = (np.float32(20.0) * np.log10(mag)).astype(np.float32)
    norm = np.clip((np.float32(24.0) - dB) / np.float32(48.0), 0, 1).astype(np.float32)
    norm = ((norm * np.float32(2) - np.float32(1)) * np.float32(-1)).astype(np.float32)
    want = np.where(norm > 0, np.exp(norm.astype(np.float64) * lnG), np.where(norm < 0, np.exp(-norm.astype(np.float64) * lnD), 1.0))
    assert np.allclose(lut, want, rtol=1e-6)
    assert lut[0] > 1.0 and lut[-1] < 1.0           # boosted bins grow, cut bins decay


def test_envelope_endpoints():
    raw = IC.raw_channels(4000, 2, 33)
    a = O.impulse_recalc(raw, stage="a")["buffers"][0]
    b = O.impulse_recalc(raw, attack=0.25, decay=0.5)["buffers"][0]
    assert b[0] == 0.0                                # attack ramp starts at 0 (Impulse.cpp:659-660)
    assert np.array_equal(b[1000:2000], np.clip(a[1000:2000], -1, 1))
    assert b[2000] == np.clip(a[2000], -1, 1)         # decay ramp starts at gain 1 (t = 0)
    assert abs(b[-1]) <= abs(a[-1]) * 0.02            # ... and ends near 0


# ---- host-side product pieces that need no device ------------------------------------------

def test_product_decay_lut_matches_oracle():
    import reevr_amd
    for key, srate, rate in (("tilt", 48000.0, 1.0), ("boost", 96000.0, 0.5), ("flat", 44100.0, 2.0)):
        got = reevr_amd.Impulse.decay_lut(IC.MAGS[key], srate, rate)
        assert np.array_equal(got, O.impulse_decay_lut(IC.MAGS[key], srate, rate))


def test_impulse_handle_without_gpu():
    import reevr_amd
    from reevr_amd import _lib
    lib = _lib.lib()
    if lib.rvc_device_count() > 0:
        pytest.skip("a GPU is present")
    imp = reevr_amd.Impulse()
    with pytest.raises(reevr_amd.RvcError):
        imp.setRaw(np.ones(10, np.float32), np.ones(10, np.float32))    # no CPU fallback
    assert lib.rvc_impulse_last_error(imp._h) == _lib.RVC_ERR_NO_DEVICE
    assert imp.size == 0
    with pytest.raises(ValueError):
        imp.setRaw(np.ones(10, np.float32))                              # 2 or 4 channels only
    # null handles are inert, like the convolver entry points
    assert lib.rvc_impulse_size(None) == 0 and lib.rvc_impulse_recalc(None, None) == 0
    lib.rvc_impulse_destroy(None)


def test_tail_start_rule():
    import reevr_amd
    a = np.zeros(100, np.float32); b = np.zeros(100, np.float32)
    assert reevr_amd.Impulse.tail_start([a, b]) == 0
    a[10] = 0.5; b[40] = -1e-3; b[60] = 9.9e-4           # 1e-3 counts, 9.9e-4 does not (Impulse.cpp:698)
    assert reevr_amd.Impulse.tail_start([a, b]) == 41
