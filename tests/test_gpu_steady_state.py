"""GPU parity at the benchmark's OWN plans, past the start-up transient (-m gpu).

bench.py measures lock-step sets in steady state: every delay line full, first-level tiles of 32 blocks, second-level
sweeps, every patch depth, child sets, the double `dif2` tail inverse, the streaming cache policy. The tests here run the
same plans -- asserted equal to bench.HEADLINE_PLANS, the table bench.py itself checks its sets against -- long enough that
at least one WHOLE first-level tile lies entirely behind the point where the delay line is full, and compare channels of
every child set sample by sample with the pinned oracle (TwoStageFFTConvolver.cpp:151-233, FFTConvolver.cpp:155-212).
Channels that share an impulse response and an input must agree bit for bit wherever in the set they run (checked on the
device). Tolerance: 1e-5 RMS relative to the output RMS (north_star), measured 2-3e-7.
"""
import importlib.util
import os

import numpy as np
import pytest

import reevr_amd
from oracle import oracle_py as O
from reevr_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def assert_bench_plan(s, cfg):
    """the set runs what bench.py's set of this configuration runs (everything but the channel count)"""
    want = _bench().HEADLINE_PLANS[cfg]
    got = s.plan()
    diff = {k: (got.get(k), v) for k, v in want.items() if got.get(k) != v}
    assert not diff, f"config {cfg}: plan differs from bench.HEADLINE_PLANS (got, want): {diff}"


def run_lockstep(nch, cfg, irs, xs, period, nblk, host_block, tail, subsets, check, tune=None, stamped=False):
    """nch lock-step channels, channel c = (irs[c % period], xs[c % period]), one process() per host block through the device
    entry; returns the output rows of `check` (host) after asserting that every channel equals its twin c % period bit for bit"""
    import torch
    full_irs = [irs[c % period] for c in range(nch)]
    dx = torch.from_numpy(np.stack(xs)).cuda().repeat(nch // period, 1)     # row c = xs[c % period]
    # (what the engine decides by the SIZE of the set is pinned to the bench's plan: the child sets, the zero-latency stage's third level)
    by_size = dict(subsets=subsets)
    if tune is None:
        by_size["head_third"] = _bench().HEADLINE_PLANS[cfg].get("head_third_level", 0)
    s = reevr_amd.ConvolverSet(nch, tune=dict(tune or {}, **by_size))
    ok = s.init(host_block, tail, full_irs, max_len=host_block) if tail else s.init_uniform(host_block, full_irs, max_len=host_block)
    assert ok, s.last_error_string
    if tune is None:
        assert_bench_plan(s, cfg)
    if stamped:                                           # (the loop with a completion stamp behind every call: same samples)
        dy, done = s.process_device_blocks_stamped(dx, host_block)
        assert len(done) == nblk and np.all(np.diff(done) >= 0) and done[0] > 0
    else:
        dy = s.process_device_blocks(dx, host_block)
    assert s.last_error == 0, s.last_error_string
    s_plan = s.plan()
    s.close()
    assert bool(torch.isfinite(dy).all())
    # twins: bit for bit inside one child set's phase group (children / phase groups run their tiles out of phase: the partial
    # sums of a channel are associated by its group's phase -- agreement to the last bits across groups, checked against the oracle)
    pl = s_plan
    parts = subsets * max(1, pl["tail_phase_groups"])
    if nch % (parts * period) == 0 and (pl["tail_phase_groups"] > 1 or (tune or {}).get("kid_stagger", 0) > 0):
        twins = dy.view(parts, nch // parts // period, period, -1)
        assert bool((twins == twins[:, 0:1]).all()), "channels with the same IR and input in one phase group differ"
        first = twins[:, 0].double()
        dev = float(((first - first[0:1]) ** 2).mean().sqrt() / (first[0:1] ** 2).mean().sqrt())
        assert dev <= 2e-6, dev
    else:
        twins = dy.view(nch // period, period, -1)
        assert bool((twins == twins[0:1]).all()), "channels with the same IR and input differ"
    got = {c: dy[c].cpu().numpy() for c in check}
    del dx, dy, twins
    torch.cuda.empty_cache()
    return got


def test_config2_plan_steady_state_vs_oracle():
    """BASELINE config 2 as bench.py runs it -- head 512 / tail 8192 one block late over IR[T,..): 16 + 58 partitions, tail tiles
    of 32 blocks (LDS-fed three-product sweep, three second-level sweeps, patches of 1..7 partitions), two child sets, tail inverse
    in double as two half-size sub-transforms, streaming sweeps (3.9 GB of tail spectra) -- at 1024 channels for 1600 blocks = 100
    tail blocks: the delay line is full from tail block 59 on, the first-level tile over tail blocks 65..96 and everything inside it
    runs on a full line. Seven channels (both children, first / last of each) against the oracle."""
    nch, head, tail, nblk, period = 1024, 512, 8192, 1600, 8
    base = [synth.synth_ir(480000, 2, inst=i) for i in range(4)]
    irs = [base[(c // 2) % 4][c % 2] for c in range(period)]
    xs = [synth.synth_input(head * nblk, 90 + i) for i in range(period)]
    check = (0, 1, 6, 511, 512, 777, 1023)
    got = run_lockstep(nch, 2, irs, xs, period, nblk, head, tail, 2, check)
    for c in check:
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c % period])
        want = o.process(xs[c % period])
        assert rel_rms(got[c], want) <= TOL, c
        # ... and over the last first-level tile alone (the steady state must not hide behind the louder start)
        lo = 65 * tail
        assert rel_rms(got[c][lo:], want[lo:]) <= TOL, c


@pytest.mark.parametrize("spread,stagger,phases", [(1, 0, 1), (3, 1, 1), (2, 1, 1), (0, 0, 8), (0, 0, 4), (1, 0, 8)])
def test_config2_geometry_spread_tail_sweeps_vs_oracle(spread, stagger, phases):
    """The same set with the tail stage's sweeps SPREAD (knob tail_spread: bit 0 the 32-block first-level sweeps, bit 1 the
    second-level ones: issued a tail period early in 15 channel slices behind the per-block launches, their newest row left to
    the patches) and the two children's tiles out of phase (kid_stagger), or -- tail_phases -- its tiles run in 8 / 4 channel groups
    out of phase with each other: no call carries a sweep over the whole set -- the reference's reason for its background
    thread, Convolver.cpp:84-95. 512 channels, through the stamped loop, against the oracle."""
    nch, head, tail, nblk, period = 512, 512, 8192, 1600, 8
    base = [synth.synth_ir(480000, 2, inst=i) for i in range(4)]
    irs = [base[(c // 2) % 4][c % 2] for c in range(period)]
    xs = [synth.synth_input(head * nblk, 90 + i) for i in range(period)]
    check = (0, 1, 255, 256, 300, 511)
    got = run_lockstep(nch, 2, irs, xs, period, nblk, head, tail, 2, check, tune=dict(tail_spread=spread, kid_stagger=stagger, tail_phases=phases), stamped=True)
    for c in check:
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c % period])
        want = o.process(xs[c % period])
        assert rel_rms(got[c], want) <= TOL, c
        assert rel_rms(got[c][65 * tail:], want[65 * tail:]) <= TOL, c


def test_config1_plan_steady_state_vs_oracle():
    """BASELINE config 1 as bench.py runs it: ONE FFTConvolver of block 512 per channel, 94 partitions, tiles 32 / 8, 8192
    channels as four child sets; 224 blocks: full from block 94 on, the tiles over blocks 128..159 and 160..191 run on a full line."""
    nch, block, nblk, period = 8192, 512, 224, 8
    irs = [synth.synth_ir(48000, 2, inst=i)[i % 2] for i in range(period)]
    xs = [synth.synth_input(block * nblk, 40 + i) for i in range(period)]
    check = (0, 3, 2047, 2048, 4099, 6143, 6150, 8191)
    got = run_lockstep(nch, 1, irs, xs, period, nblk, block, 0, 4, check)
    for c in check:
        o = O.FFTConvolver("orc")
        assert o.init(block, irs[c % period])
        want = o.process(xs[c % period])
        assert rel_rms(got[c], want) <= TOL, c
        assert rel_rms(got[c][128 * block:], want[128 * block:]) <= TOL, c


def test_config5_geometry_plan_steady_state_vs_oracle():
    """BASELINE config 5's geometry in the lock-step regime as bench.py runs it: head 4096 / tail 8192 one block late, 2 + 29
    partitions, per-block calls through the general path (transform / delay line / inverse launches), BOTH stages' inverse
    transforms in double, tail tiles of 16 blocks; 256 channels as two child sets for 64 tail blocks: full from tail block 30 on,
    the tile over tail blocks 33..48 runs on a full line."""
    nch, block, tail, period = 256, 4096, 8192, 8
    nblk = 64 * (tail // block)
    base = [synth.synth_ir(240000, 2, inst=10 + i) for i in range(4)]
    irs = [base[(c // 2) % 4][c % 2] for c in range(period)]
    xs = [synth.synth_input(block * nblk, 60 + i) for i in range(period)]
    check = (0, 5, 127, 128, 255)
    got = run_lockstep(nch, 5, irs, xs, period, nblk, block, tail, 2, check)
    for c in check:
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(block, tail, irs[c % period])
        want = o.process(xs[c % period])
        assert rel_rms(got[c], want) <= TOL, c
        assert rel_rms(got[c][33 * tail:], want[33 * tail:]) <= TOL, c


def test_config3_plan_is_the_benchs():
    """config 3's geometry is compared with the oracle over all of its 64 + 350 / 175 partitions in
    test_gpu_parity.py::test_two_level_tiling_at_config3_geometry (64 channels, tiling forced); here: the DEFAULT set of 256
    channels takes the plan bench.py's 2048-channel set runs (tail at block 16384 one block late, tiles 32 / 32, float
    transforms), and 40 blocks of it are finite and agree with the oracle on two channels."""
    import torch
    nch, head, tail, nblk = 256, 256, 8192, 40
    base = synth.synth_ir(2880000, 2, 0)
    irs = [base[c % 2] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 200 + c % 2) for c in range(nch)])
    s = reevr_amd.ConvolverSet(nch, tune=dict(subsets=2))
    assert s.init(head, tail, irs, max_len=head), s.last_error_string
    assert_bench_plan(s, 3)
    got = s.process_device_blocks(torch.from_numpy(x).cuda(), head).cpu().numpy()
    assert s.last_error == 0, s.last_error_string
    s.close()
    for c in (0, nch - 1):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        assert rel_rms(got[c], o.process(x[c])) <= TOL, c


def test_lockstep_4096_channels_impulse_identity_past_the_ir():
    """bench.py's default set itself (4096 lock-step channels, 33 GB resident, two child sets: the plan is asserted) for 960
    blocks = 491 520 samples, past the END of the 480 000-sample impulse responses: a unit impulse in gives the IR out -- every
    one of the 16 + 58 partitions, every tile level -- and then silence; channels that share an IR agree bit for bit."""
    import torch
    nch, head, tail, nblk, at = 4096, 512, 8192, 960, 5
    base = [synth.synth_ir(480000, 2, inst=i) for i in range(4)]
    irs = [base[(c // 2) % 4][c % 2] for c in range(nch)]
    s = reevr_amd.ConvolverSet(nch)
    assert s.init(head, tail, irs, max_len=head), s.last_error_string
    assert_bench_plan(s, 2)
    assert s.plan()["channels"] == _bench().WORKLOADS[2]["channels"]
    n = head * nblk
    dx = torch.zeros((nch, n), device="cuda")
    dx[:, at] = 1.0
    dy = s.process_device_blocks(dx, head)
    assert s.last_error == 0, s.last_error_string
    s_subsets, s_phases = s.subsets, max(1, s.plan()["tail_phase_groups"])
    s.close()
    for c in range(8):                                   # 8 distinct (IR, channel) combinations, cycled
        ref = dy[c]
        want = np.zeros(n, np.float32)
        want[at:at + 480000] = irs[c]
        got = ref.cpu().numpy().astype(np.float64)
        err = np.sqrt(np.mean((got - want) ** 2))
        assert err <= 1e-7, (c, err)
        assert np.abs(got[at + 480000:]).max() <= 1e-6, c            # silence behind the impulse response
        # the last tail partition alone (the part of the IR only a FULL-length run reaches)
        lo = at + 57 * tail + tail
        assert np.sqrt(np.mean((got[lo:] - want[lo:]) ** 2)) <= 1e-7, c
        # channels c, c + 8, ..: bit for bit inside one phase group of one child set (each group runs its tiles on its own phase and
        # associates its partial sums accordingly), every group against the impulse response
        groups = s_subsets * s_phases
        same = dy[c::8].view(groups, nch // 8 // groups, n)
        assert bool((same == same[:, 0:1]).all()), c
        heads = same[:, 0].cpu().numpy().astype(np.float64)
        for g in range(groups):
            assert np.sqrt(np.mean((heads[g] - want) ** 2)) <= 1e-7, (c, g)
