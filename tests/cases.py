"""Shared parity-case definitions (used by oracle/gen_golden.py and the tests).

KAT_* are the 58 known-answer tuples of the reference's own test program
(libs/FFTConvolver/test/Test.cpp:256-288 and :297-329): ramp input, ramp IR, the
convolver fed in random-sized calls and flushed with zeros until inputSize+irSize-1
outputs exist. The reference draws call sizes from unseeded glibc rand(); here the
schedule comes from a seeded numpy generator so it is reproducible everywhere (the
reference's result is call-pattern independent to ~2e-8 RMS).

SYNTH_CASES are the BASELINE.json configurations (shortened inputs) and the edge cases
the reference's semantics define: IR shorter than one tail block, between one and two,
trailing zeros, all-zero IR, block-aligned clear(), ragged call sizes.
"""
from __future__ import annotations

import numpy as np

# (inputSize, irSize, blockSizeMin, blockSizeMax, blockSizeConvolver)   Test.cpp:256-288
KAT_FFTCONV = [
    (1, 1, 1, 1, 1), (2, 2, 2, 2, 2), (3, 3, 3, 3, 3),
    (3, 2, 2, 2, 2), (4, 2, 2, 2, 2), (4, 3, 2, 2, 2), (9, 4, 3, 3, 2), (171, 7, 5, 5, 5),
    (1979, 17, 7, 7, 5), (100, 10, 3, 5, 5), (123, 45, 12, 34, 34),
    (2, 3, 2, 2, 2), (2, 4, 2, 2, 2), (3, 4, 2, 2, 2), (4, 9, 3, 3, 3), (7, 171, 5, 5, 5),
    (17, 1979, 7, 7, 7), (10, 100, 3, 5, 5), (45, 123, 12, 34, 34),
    (100000, 1234, 100, 128, 128), (100000, 1234, 100, 256, 256), (100000, 1234, 100, 512, 512),
    (100000, 1234, 100, 1024, 1024), (100000, 1234, 100, 2048, 2048),
    (100000, 4321, 100, 128, 128), (100000, 4321, 100, 256, 256), (100000, 4321, 100, 512, 512),
    (100000, 4321, 100, 1024, 1024), (100000, 4321, 100, 2048, 2048),
]

# (inputSize, irSize, blockSizeMin, blockSizeMax, blockSizeHead, blockSizeTail)  Test.cpp:297-329
KAT_TWOSTAGE = [
    (1, 1, 1, 1, 1, 1), (2, 2, 2, 2, 2, 2), (3, 3, 3, 3, 3, 3),
    (3, 2, 2, 2, 2, 4), (4, 2, 2, 2, 2, 4), (4, 3, 2, 2, 2, 4), (9, 4, 3, 3, 2, 4),
    (171, 7, 5, 5, 5, 10), (1979, 17, 7, 7, 5, 10), (100, 10, 3, 5, 5, 10), (123, 45, 12, 34, 34, 68),
    (2, 3, 2, 2, 1, 2), (2, 4, 2, 2, 1, 2), (3, 4, 2, 2, 1, 2), (4, 9, 3, 3, 2, 4),
    (7, 171, 5, 5, 2, 16), (17, 1979, 7, 7, 4, 16), (10, 100, 3, 5, 1, 4), (45, 123, 12, 34, 4, 32),
    (100000, 1234, 100, 128, 128, 4096), (100000, 1234, 100, 256, 256, 4096),
    (100000, 1234, 100, 512, 512, 4096), (100000, 1234, 100, 1024, 1024, 4096),
    (100000, 1234, 100, 2048, 2048, 4096),
    (100000, 4321, 100, 128, 128, 4096), (100000, 4321, 100, 256, 256, 4096),
    (100000, 4321, 100, 512, 512, 4096), (100000, 4321, 100, 1024, 1024, 4096),
    (100000, 4321, 100, 2048, 2048, 4096),
]


def kat_name(kind: str, tup) -> str:
    return kind + "_" + "_".join(str(v) for v in tup)


def schedule(total: int, lo: int, hi: int, seed: int) -> list:
    """Call sizes in [lo, hi] covering exactly `total` samples (Test.cpp:104-124)."""
    rng = np.random.RandomState(seed)
    out, done = [], 0
    while done < total:
        n = lo + int(rng.randint(0, 1 + hi - lo))
        n = min(n, total - done)
        out.append(n)
        done += n
    return out


def kat_tolerance_ok(out: np.ndarray, exact: np.ndarray, ir_len: int) -> bool:
    """The reference's pass rule (Test.cpp:129-145), `exact` being the direct convolution."""
    a = out.astype(np.float64)
    b = exact.astype(np.float64)
    m = (np.abs(a) > 1.0) & (np.abs(b) > 1.0)
    abs_err = np.abs(a - b)[m]
    rel_err = abs_err / b[m]
    bad = (rel_err > 1e-4 * np.log(float(ir_len))) & (abs_err > 1e-3 * float(ir_len))
    return int(bad.sum()) == 0


def kat_margin(out: np.ndarray, exact: np.ndarray, ir_len: int) -> float:
    """How close the worst sample comes to failing the rule above: a sample fails when BOTH its
    relative and absolute error exceed their tolerance, so its margin is the smaller of the two
    ratios; < 1 passes. (The reference itself scores <= 0.07 on its 58 cases.)"""
    a = out.astype(np.float64)
    b = exact.astype(np.float64)
    m = (np.abs(a) > 1.0) & (np.abs(b) > 1.0)
    if not m.any():
        return 0.0
    abs_err = np.abs(a - b)[m]
    rel_err = abs_err / b[m]
    return float(np.minimum(abs_err / (1e-3 * float(ir_len)), rel_err / (1e-4 * np.log(float(ir_len)))).max())


# ---- synthetic cases ---------------------------------------------------------------
# kind: "fftconv" (block) or "twostage" (head, tail)
# ir:   ("synth", irLen, nChannels, inst) | ("zeros", irLen) | ("synth_trailing_zeros", irLen, nzeros)
# calls: ("fixed", n) | ("ragged", lo, hi, seed)
# clear_at: list of frame positions (multiples of the head block) where clear() is called
SYNTH_CASES = {
    # BASELINE.json configs[0]: mono, 1 s IR @ 48 kHz, block 512, single FFTConvolver
    "cfg1_mono_1s_b512": dict(kind="fftconv", block=512, ir=("synth", 48000, 1, 0),
                              frames=512 * 188, calls=("fixed", 512)),
    # configs[1]: stereo, 10 s IR @ 48 kHz, block 512 -> head 512 / tail 8192
    "cfg2_stereo_10s_b512": dict(kind="twostage", head=512, tail=8192, ir=("synth", 480000, 2, 0),
                                 frames=512 * 1200, calls=("fixed", 512)),
    # configs[2]: stereo, 30 s IR @ 96 kHz, block 256 -> head 256 / tail 8192
    "cfg3_stereo_30s96k_b256": dict(kind="twostage", head=256, tail=8192, ir=("synth", 2880000, 2, 0),
                                    frames=256 * 4608, calls=("fixed", 256)),
    # configs[2] at full length: input = IR length + two wraps of the tail delay line's 512-row ring
    # (352 + 1024 tail blocks), so all 350 tail partitions carry signal and every ring wraps
    "cfg3_full_length": dict(kind="twostage", head=256, tail=8192, ir=("synth", 2880000, 2, 0),
                             frames=8192 * 1376, calls=("fixed", 256)),
    # configs[3]: one of the 8 independent stereo instances (inst 3), 10 s IR @ 48 kHz
    "cfg4_inst3_10s_b512": dict(kind="twostage", head=512, tail=8192, ir=("synth", 480000, 2, 3),
                                frames=512 * 400, calls=("fixed", 512)),
    # configs[4]: one channel pair of the 64-channel offline render, 5 s IR, block 4096
    "cfg5_5s_b4096": dict(kind="twostage", head=4096, tail=8192, ir=("synth", 240000, 2, 0),
                          frames=4096 * 100, calls=("fixed", 4096)),
    # small geometries that exercise all three sub-convolvers / only one / only two
    "small_three_stage": dict(kind="twostage", head=32, tail=128, ir=("synth", 1000, 2, 5),
                              frames=5000, calls=("ragged", 1, 97, 11)),
    "small_ir_le_tail": dict(kind="twostage", head=32, tail=128, ir=("synth", 100, 1, 6),
                             frames=3000, calls=("ragged", 1, 64, 12)),
    "small_ir_le_2tail": dict(kind="twostage", head=32, tail=128, ir=("synth", 200, 1, 7),
                              frames=3000, calls=("ragged", 1, 64, 13)),
    "small_ir_eq_2tail": dict(kind="twostage", head=32, tail=128, ir=("synth", 256, 1, 8),
                              frames=3000, calls=("ragged", 1, 300, 14)),
    "nonpow2_sizes": dict(kind="twostage", head=24, tail=100, ir=("synth", 777, 1, 9),
                          frames=4000, calls=("ragged", 1, 50, 15)),
    "head_eq_tail": dict(kind="twostage", head=64, tail=64, ir=("synth", 500, 1, 10),
                         frames=3000, calls=("ragged", 1, 200, 16)),
    "ragged_calls_b512": dict(kind="twostage", head=512, tail=8192, ir=("synth", 40000, 2, 11),
                              frames=60000, calls=("ragged", 1, 1500, 17)),
    "large_calls_b512": dict(kind="twostage", head=512, tail=8192, ir=("synth", 40000, 1, 12),
                             frames=100000, calls=("ragged", 5000, 30000, 18)),
    "fftconv_ragged": dict(kind="fftconv", block=128, ir=("synth", 3000, 1, 13),
                           frames=20000, calls=("ragged", 1, 700, 19)),
    "zero_ir": dict(kind="twostage", head=64, tail=256, ir=("zeros", 1000),
                    frames=1024, calls=("fixed", 64)),
    "trailing_zero_ir": dict(kind="twostage", head=64, tail=256, ir=("synth_trailing_zeros", 900, 300),
                             frames=4000, calls=("ragged", 1, 200, 20)),
    "clear_block_aligned": dict(kind="twostage", head=64, tail=256, ir=("synth", 1500, 2, 14),
                                frames=64 * 120, calls=("fixed", 64), clear_at=[64 * 37, 64 * 80]),
}


def make_ir(spec) -> np.ndarray:
    """(nChannels, irLen) float32."""
    from reevr_amd import synth
    if spec[0] == "synth":
        return synth.synth_ir(spec[1], spec[2], spec[3])
    if spec[0] == "zeros":
        return np.zeros((1, spec[1]), np.float32)
    if spec[0] == "synth_trailing_zeros":
        ir = synth.synth_ir(spec[1], 1, 21)
        ir[:, spec[1] - spec[2]:] = 0.0
        ir[:, spec[1] - spec[2] + 5] = 5e-7  # below the 1e-6 trim threshold (FFTConvolver.cpp:102-106)
        return ir
    raise ValueError(spec)


def make_calls(case) -> list:
    c = case["calls"]
    if c[0] == "fixed":
        assert case["frames"] % c[1] == 0
        return [c[1]] * (case["frames"] // c[1])
    return schedule(case["frames"], c[1], c[2], c[3])


def make_input(case, n_channels: int) -> np.ndarray:
    from reevr_amd import synth
    return np.stack([synth.synth_input(case["frames"], c) for c in range(n_channels)])


def decimate_idx(n: int, max_pts: int = 4096) -> np.ndarray:
    if n <= max_pts:
        return np.arange(n)
    return np.unique(np.linspace(0, n - 1, max_pts).astype(np.int64))


# ---- runners: `factory(kind)` returns an object with init(...)/process(x)/clear() ------
def run_kat(factory, kind: str, tup) -> np.ndarray:
    from reevr_amd import synth
    if kind == "fftconv":
        n_in, n_ir, lo, hi, blk = tup
        conv = factory("fftconv")
        conv.init(blk, synth.ramp(n_ir))
    else:
        n_in, n_ir, lo, hi, head, tail = tup
        conv = factory("twostage")
        conv.init(head, tail, synth.ramp(n_ir))
    total = n_in + n_ir - 1
    x = np.zeros(total, np.float32)
    x[:n_in] = synth.ramp(n_in)
    seed = (n_in * 31 + n_ir * 17 + hi) & 0x7FFFFFFF
    out = np.empty(total, np.float32)
    pos = 0
    for n in schedule(total, lo, hi, seed):
        out[pos:pos + n] = conv.process(x[pos:pos + n])
        pos += n
    return out


def run_synth_case(factory, case) -> np.ndarray:
    """Returns (nChannels, frames) float32: channel c = IR channel c applied to input c."""
    irs = make_ir(case["ir"])
    nch = irs.shape[0]
    x = make_input(case, nch)
    calls = make_calls(case)
    clear_at = set(case.get("clear_at", []))
    outs = []
    for c in range(nch):
        conv = factory(case["kind"])
        if case["kind"] == "fftconv":
            assert conv.init(case["block"], irs[c])
        else:
            assert conv.init(case["head"], case["tail"], irs[c])
        out = np.empty(case["frames"], np.float32)
        pos = 0
        for n in calls:
            if pos in clear_at:
                conv.clear()
            out[pos:pos + n] = conv.process(x[c, pos:pos + n])
            pos += n
        outs.append(out)
    return np.stack(outs)


def summarize(out: np.ndarray) -> dict:
    """Small fixture of a 1-D output: decimated samples + head/tail + rms + sum."""
    o64 = out.astype(np.float64)
    idx = decimate_idx(out.size)
    return dict(n=np.int64(out.size), dec=out[idx].copy(), head=out[:512].copy(),
                tail=out[-512:].copy(), rms=np.float64(np.sqrt(np.mean(o64 ** 2))) if out.size else np.float64(0),
                sum=np.float64(o64.sum()))


def compare_to_fixture(out: np.ndarray, fx: dict, rel_rms_tol: float) -> float:
    """Max over (dec, head, tail) of RMS error relative to the fixture's RMS; asserts."""
    assert int(fx["n"]) == out.size
    idx = decimate_idx(out.size)
    scale = float(fx["rms"]) if float(fx["rms"]) > 0 else 1.0
    worst = 0.0
    for got, want in ((out[idx], fx["dec"]), (out[:512], fx["head"]), (out[-512:], fx["tail"])):
        d = got.astype(np.float64) - want.astype(np.float64)
        err = float(np.sqrt(np.mean(d ** 2))) / scale if d.size else 0.0
        worst = max(worst, err)
    assert worst <= rel_rms_tol, f"rel RMS error {worst:.3e} > {rel_rms_tol:.1e}"
    return worst
