"""CPU tests of the host-side logic that needs no GPU: the synthetic-input generators, bench.py's
algorithmic byte model (against SURVEY.md 8d's figures), and the IR hot-swap state machine
(reevr_amd/hotswap.py) driven by oracle convolvers against the oracle-side restatement of the
reference sequence."""
import importlib.util
import os

import numpy as np
import pytest

from reevr_amd import synth
from reevr_amd.hotswap import HotSwapStereoConvolver, wet_bus
from tests.ref_hotswap import OracleStereoConvolver, RefHotSwap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_synth_is_counter_based_and_deterministic():
    a = synth.white_noise(1000, 123)
    b = np.concatenate([synth.white_noise(400, 123), synth.white_noise(600, 123, offset=400)])
    assert np.array_equal(a, b)                       # chunking invariant
    assert a.dtype == np.float32 and a.min() >= -1.0 and a.max() < 1.0
    assert abs(float(a.std()) - 0.577) < 0.03         # uniform [-1, 1)
    assert not np.array_equal(a, synth.white_noise(1000, 124))
    ir = synth.synth_ir(48000, 2, 0)
    e = float(np.sum(ir.astype(np.float64) ** 2))
    assert abs(e - 1.0) < 1e-3                        # calculateAutoGain: unit energy over L+R (Impulse.cpp:691-708)
    assert abs(ir[0, -1]) < 2e-3 * np.abs(ir[0]).max() * 10   # -60 dB at the end


def test_algorithmic_byte_model_matches_survey():
    """1531 / 1497 / 6701 / 584 B per channel-sample for configs 1 / 2,4 / 3 / 5."""
    b = _bench()
    assert b.alg_bytes_block(512, 94) == 783848
    assert round(b.alg_bytes_block(512, 94) / 512) == 1531
    assert round(b.alg_bytes_per_sample(512, 8192, 480000)) == 1497
    assert round(b.alg_bytes_per_sample(256, 8192, 2880000)) == 6701
    assert round(b.alg_bytes_per_sample(4096, 8192, 240000)) == 584


def test_bench_traffic_lookup_is_keyed_on_the_launch_size():
    """bench.py's counter-measured bytes per launch apply only to launches of exactly the channel count they were measured with
    (round 3's line printed half-size figures for configs 1 / 3): the committed profiles/r6_traffic.json serves the default run
    (child sets: 2048 / 2048 / 1024 / 2048 channels per launch for configs 2 / 1 / 3 / 5) and the one-queue run (all the
    channels per launch), and any other launch size gets nothing."""
    b = _bench()
    for cfg, per_launch, one_queue in ((2, 2048, 4096), (1, 2048, 8192), (3, 1024, 2048), (5, 2048, 4096)):
        t, src = b.load_traffic(per_launch, cfg, True)
        assert t and "r6_traffic.json" in src and str(per_launch) in src
        t1, _ = b.load_traffic(one_queue, cfg, True)
        assert t1 and set(t1) == set(t)
        for k in t:                                   # twice the channels per launch: twice the bytes, within a few percent
            assert 0.9 < t1[k] / (t[k] * one_queue / per_launch) < 1.1, (cfg, k)
        assert b.load_traffic(per_launch + 2, cfg, True) == ({}, None)
        assert b.load_traffic(per_launch, cfg, False) == ({}, None)       # (measured with the time tiling on)


def test_stage_plan_policy():
    """rvc_debug_plan (the pure function rvc_set_init applies, rvc_plan.cpp plan_stages): the reference's structure -- zero-latency
    stage over IR[0,2T), tail at block T two blocks late (TwoStageFFTConvolver.cpp:117-138, :213-222) -- for small sets, sets with
    the tail on a second stream, fixed partitions or the reference-order schedule; for lock-step sets of >= 256 channels the tail one
    block late: at block 2T for tails of >= 128 partitions (>= 48 when 2T < 16384), else over IR[T,..) with half the zero-latency
    stage. Partition counts of the three BASELINE geometries as bench.py measures them."""
    import reevr_amd
    from reevr_amd import _lib as L
    P = reevr_amd.stage_plan
    ref2 = {"head_block": 512, "tail_block": 8192, "zero_latency_samples": 16384, "tail_delay": 2, "partitions": (32, 57)}
    assert P(2, 512, 8192, 480000) == ref2 and P(255, 512, 8192, 480000) == ref2            # the plug-in's pair; below the threshold
    for flag in (L.RVC_FLAG_BG_STREAM, L.RVC_FLAG_FIXED_PARTITIONS, L.RVC_FLAG_NO_TIME_TILING):
        assert P(4096, 512, 8192, 480000, flag) == ref2
    assert P(4096, 512, 8192, 480000) == dict(ref2, zero_latency_samples=8192, tail_delay=1, partitions=(16, 58))      # config 2: shrunk
    assert P(2048, 256, 8192, 2880000) == {"head_block": 256, "tail_block": 16384, "zero_latency_samples": 16384, "tail_delay": 1,
                                           "partitions": (64, 175)}                                                 # config 3: widened
    assert P(4096, 4096, 8192, 240000)["partitions"] == (2, 29)                                                       # config 5's geometry
    assert P(2048, 256, 8192, 2880000, L.RVC_FLAG_FFT_F64_LONG)["partitions"] == (32, 351)    # double tail transforms: no 16384-bin form
    assert P(2048, 256, 8192, 2880000, L.RVC_FLAG_FFT_F64)["tail_block"] == 8192
    assert P(4096, 512, 8192, 2 * 8192 + 127 * 8192)["tail_block"] == 8192 and P(4096, 512, 8192, 2 * 8192 + 128 * 8192)["tail_block"] == 16384
    assert P(4096, 256, 2048, 2 * 2048 + 48 * 2048)["tail_block"] == 4096 and P(4096, 256, 2048, 2 * 2048 + 47 * 2048)["tail_block"] == 2048
    assert P(4096, 512, 8192, 16384) == dict(ref2, partitions=(32, 0))        # no tail at the reference's geometry: nothing to move
    assert P(4096, 8192, 8192, 480000)["tail_delay"] == 2                     # head = tail: there is no zero-latency stage to shrink
    assert P(4096, 8192, 512, 480000) == P(4096, 512, 8192, 480000)           # head > tail: swapped like the reference (:100-104)
    with reevr_amd.tuning(tail_slack=0):
        assert P(4096, 512, 8192, 480000) == ref2
    with reevr_amd.tuning(tail_slack=1):
        assert P(2, 512, 8192, 480000, L.RVC_FLAG_FFT_F32)["tail_block"] == 16384 and P(2, 512, 8192, 480000)["tail_block"] == 8192
    with pytest.raises(ValueError):
        P(0, 512, 8192, 1000)


class _SetGeometry:
    """what bench.executed_bytes asks a ConvolverSet for"""

    def __init__(self, parts, tiles, tail_block, subsets=1, patch_in_launch=0, phases=1, third=(0, 0)):
        self._p, self._t, self.tail_block, self.subsets, self._pil, self._ph, self._third = parts, tiles, tail_block, subsets, patch_in_launch, phases, third

    def plan(self):
        return {"head_patch_in_launch": self._pil, "tail_phase_groups": self._ph, "tail_spread": 0, "tail_sweep_slices": 1,
                "head_third_level": self._third[0], "tail_third_level": self._third[1]}

    def partitions(self, stage):
        return self._p[stage]

    def tile_rows(self, stage):
        return self._t[stage]


def test_executed_bytes_model_against_the_committed_counter_passes():
    """bench.py's executed-bytes model of every kernel family -- with the structures the engine runs for sets of thousands of
    channels: the tail one block late over IR[T,..) and half the zero-latency stage (configs 2 / 5: 16 + 58 and 2 + 29 partitions),
    the tail at block 16384 (config 3: 64 + 175) -- against the PMC bytes per launch of the committed one-queue passes
    (profiles/r6_traffic.json, rocprofv3 FETCH_SIZE / WRITE_SIZE): within 5 % for every family, none missing. Round 6: the tail tiles run
    in 8 channel groups out of phase -- every sweep / patch launch of the tail stage covers an eighth of the channels. Round 5: the per-block
    launch of configs 1 / 2 / 3 (heads 512 / 512 / 256) patches its own block and hands the row over through LDS (no accumulator
    round trip through memory); config 5's head of 4096 keeps the general per-block path. Round 6, later: third-level sweeps (tail: all
    four; zero-latency stage: configs 1 / 2) and the zero-latency stage's sweeps taking the newest row."""
    b = _bench()
    cases = {2: (4096, 512, 8192, 480000, _SetGeometry((16, 58), (8, 32), 8192, patch_in_launch=1, phases=8, third=(1, 1))),
             3: (2048, 256, 8192, 2880000, _SetGeometry((64, 175), (32, 32), 16384, patch_in_launch=1, phases=8, third=(0, 1))),
             5: (4096, 4096, 8192, 240000, _SetGeometry((2, 29), (0, 16), 8192, phases=8, third=(0, 1))),
             1: (8192, 512, 0, 48000, _SetGeometry((94, 0), (32, 0), 0, patch_in_launch=1, third=(1, 0)))}
    for cfg, (nch, head, tail, ir_len, conv) in cases.items():
        exe = b.executed_bytes(conv, nch, head, tail, ir_len, head, True)
        traffic, src = b.load_traffic(nch, cfg, True)
        assert traffic, cfg
        assert b.tail_stage_form(conv, head, tail).startswith({2: "delay 1, block T", 3: "delay 1, block 2T", 5: "delay 1, block T", 1: "none"}[cfg])
        for fam, measured in traffic.items():
            assert fam in exe, (cfg, fam)
            assert 0.95 <= measured / exe[fam] <= 1.05, (cfg, fam, measured / exe[fam])


class _Imp:
    pass


def _imp(inst, n, quad):
    irs = synth.synth_ir(n, 4, inst)
    m = _Imp()
    m.bufferLL, m.bufferRR, m.bufferLR, m.bufferRL = irs
    m.isQuad = quad
    return m


@pytest.mark.parametrize("quad", [False, True])
def test_hot_swap_state_machine_on_cpu(quad):
    """hotswap.HotSwapStereoConvolver (vectorised crossfade, warm-up through warm()/fallback loop)
    == the literal restatement of src/PluginProcessor.cpp:1655-1756, 1793-1838, both on oracle
    convolvers, including the buffer-length crossfade quirk with short host blocks."""
    sr, blk, nblocks = 8000, 96, 60                   # small: 0.25 s warm ring = 2000 samples
    a, b = _imp(90, 3000, quad), _imp(91, 2500, quad)
    L = synth.synth_input(blk * nblocks, 0); R = synth.synth_input(blk * nblocks, 1)
    new = HotSwapStereoConvolver(lambda: OracleStereoConvolver(), threaded=False)
    ref = RefHotSwap()
    for h in (new, ref):
        h.prepare(sr, blk)
        h.loadImpulse(a)
    got, want = [], []
    for i in range(nblocks):
        n = blk if i % 7 else blk - 11                # hosts may deliver short blocks
        s = slice(i * blk, i * blk + n)
        if i == 25:
            assert new.request_impulse(b)
            ref.request_impulse(b)
        got.append(new.process(L[s], R[s], L[s], R[s], n))
        want.append(ref.process(L[s], R[s], L[s], R[s], n))
    got = np.concatenate(got, axis=1); want = np.concatenate(want, axis=1)
    assert new.loadState == 0 and ref.state == 0
    assert np.abs(got - want).max() <= 2e-6
    assert not new.request_impulse(b) or True


def test_wet_bus_matches_scalar_loop():
    """f-3: reverb envelope x mid/side width x dry/wet (src/PluginProcessor.cpp:1840-1876)."""
    rng = np.random.RandomState(3)
    wet = rng.randn(2, 64).astype(np.float32); dry = rng.randn(2, 64).astype(np.float32)
    yrev = rng.rand(64).astype(np.float32)
    width, dg, wg = 0.7, 0.5, 0.8
    out = wet_bus(wet, yrev, width, dg, wg, dry)                 # the product's host-side form ...
    from tests.ref_wetbus import ref_wet_bus
    want = ref_wet_bus(wet, yrev, width, dg, wg, dry)            # ... against the oracle-side scalar restatement
    assert np.array_equal(out, want)


def test_pmc_summary_kernel_families():
    """tools/pmc_summarize.py: rocprofv3 kernel names -> the families bench.py reports (first- vs second-level sweeps are told
    apart by the instantiation, not by the tile size)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "pmc_summarize", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pmc_summarize.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    fam = lambda n: m.family(n, 9, 13)
    assert fam("void rvc::k_fused_block2w<9, false, false>(rvc::FusedArgs, rvc::FirArgs)") == "fused_block"
    assert fam("void rvc::k_fdl_sweep<16, 1, 0, 2, 4, 3, true>(rvc::FirArgs, int)") == "sweep_head"
    assert fam("void rvc::k_fdl_sweep<8, 1, 0, 4, 4, 3, false>(rvc::FirArgs, int)") == "sweep2_head"
    assert fam("void rvc::k_fdl_sweep<8, 1, 0, 4, 4, 3, true>(rvc::FirArgs, int)") == "sweep_head"
    assert fam("void rvc::k_fdl_sweep<8, 4, 1, 4, 4, 3, false>(rvc::FirArgs, int)") == "sweep_tail"
    assert fam("void rvc::k_fdl_sweep<16, 1, 1, 4, 4, 2, true>(rvc::FirArgs, int)") == "sweep_tail"
    assert fam("void rvc::k_fdl_sweep<8, 1, 1, 4, 4, 3, false>(rvc::FirArgs, int)") == "sweep2_tail"
    assert fam("void rvc::k_fdl_sweep_lds<16, 2, 1, 1, true, 4>(rvc::FirArgs, int)") == "sweep_tail"
    assert fam("void rvc::k_fdl_sweep_lds<16, 2, 1, 0, true, 4>(rvc::FirArgs, int)") == "sweep_head"
    assert fam("void rvc::k_fdl_patch<1, true>(rvc::FirArgs, int)") == "fir_tail"
    assert fam("void rvc::k_fft8_inv<13, float, false>(rvc::InvArgs)") == "fft_inv_tail"
    assert fam("void rvc::k_fft8_fwd_loop<13>(rvc::FwdArgs, int)") == "fft_fwd_tail"
    assert fam("void rvc::k_fft8_fwd<13, double>(rvc::FwdArgs)") is None
    # round 5: the three-product form of the LDS-fed sweep (one more template argument), the tail inverse in double
    assert fam("void rvc::k_fdl_sweep_lds<16, 2, 1, 1, true, 4, true>(rvc::FirArgs, int)") == "sweep_tail"
    assert fam("void rvc::k_fdl_sweep_lds<16, 2, 1, 0, true, 4, false>(rvc::FirArgs, int)") == "sweep_head"
    assert fam("void rvc::k_fft8_inv<13, double, false, false>(rvc::InvArgs)") == "fft_inv_tail"
    assert fam("void rvc::k_fft8_inv_dif2<12, double>(rvc::InvArgs, int)") == "fft_inv_tail"
    # the sweeps' trailing NTH argument (IR rows non-temporal where the delay-line rows' loads are ordinary: big stages' second level)
    assert fam("void rvc::k_fdl_sweep<8, 1, 1, 4, 4, 3, false, true>(rvc::FirArgs, int)") == "sweep2_tail"
    assert fam("void rvc::k_fdl_sweep<4, 1, 1, 4, 4, 4, false, true>(rvc::FirArgs, int)") == "sweep3_tail"     # round 6: third level
    assert fam("void rvc::k_fdl_sweep<8, 1, 0, 4, 4, 3, false, false>(rvc::FirArgs, int)") == "sweep2_head"
    assert fam("void rvc::k_fdl_sweep<8, 1, 0, 4, 4, 3, true, true>(rvc::FirArgs, int)") == "sweep_head"
    assert fam("void rvc::k_fdl_sweep<16, 1, 1, 4, 4, 2, true, true>(rvc::FirArgs, int)") == "sweep_tail"



def test_every_hot_kernel_of_the_committed_summaries_has_a_family():
    """The rocprofv3 summaries kept under profiles/r6_config<C>/ list C++ kernel names; tools/pmc_summarize.py (and through it
    bench.py's `traffic`) sorts them into families by those names. Every kernel that takes more than 0.2 % of a kept run must map
    to a family -- a renamed or new kernel that silently drops out of the accounting fails here -- except the launches of init time
    (the IR spectra's double forward transforms, the runtime's fill / copy kernels)."""
    import csv
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_summarize", os.path.join(ROOT, "tools", "pmc_summarize.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    geometry = {1: (9, 13), 2: (9, 13), 3: (8, 14), 5: (12, 13)}          # (head, tail) log2 block sizes: tools/profile_configs.sh
    init_only = ("k_fft8_fwd<13, double", "k_fft8_fwd<14, double", "k_fft8_fwd<12, double", "__amd_rocclr_")
    seen = set()
    for cfg, (hl, tl) in geometry.items():
        for name in ("kernel_stats.csv", "kernel_stats_one_queue.csv"):
            rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", f"r6_config{cfg}", name))))
            total = sum(float(r["TotalDurationNs"]) for r in rows)
            for r in rows:
                if float(r["TotalDurationNs"]) / total <= 0.002 or any(t in r["Name"] for t in init_only):
                    continue
                fam = m.family(r["Name"], hl, tl)
                assert fam is not None, (cfg, name, r["Name"])
                seen.add(fam)
    assert {"fused_block", "sweep_tail", "sweep2_tail", "sweep_head", "fir_tail", "fft_fwd_tail", "fft_inv_tail"} <= seen


def test_compact_bench_line_fits_and_carries_the_contract():
    """bench.py prints ONE compact line last (the driver parses it; round 4's 24 KB line could not be parsed) and writes the
    full record to a side file: the line builder on a canned full record (round 4's, the largest so far) and on an inflated
    one must stay under 4 KB and keep every key of the contract."""
    import copy
    import json
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r4_bench.json")))
    full["config"].update(devices=[0], shared_device=False)
    fat = copy.deepcopy(full)
    fat["config"]["workload"] = "x" * 5000
    fat["cpu_baseline"]["sample"] = "y" * 5000
    fat["roofline"]["traffic_source"] = "z" * 5000
    fat["config"]["partitions"] = {"stage %d" % i: i for i in range(400)}
    for rec in (full, fat):
        line = b.compact_line(rec, "bench_full.json")
        text = json.dumps(line)
        assert len(text) < 4096, len(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline", "probe"):
            assert k in line, k
        for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in line["roofline"], k
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in line["cpu_baseline"], k
        assert line["config"]["workload"] and line["config"]["baseline_config"] == 2
        assert line["value"] == rec["value"] and line["roofline"]["frac"] == rec["roofline"]["frac"]
    # the normal record keeps the optional groups too
    line = b.compact_line(full, "bench_full.json")
    assert set(line["other_configs"]) == {"1", "3", "5"} and "ch2_us_per_block" in line["side"]
    # round 6's record: what one call costs, the plan check, the host boundary, the literal config 5's roofline ride along
    r6 = json.load(open(os.path.join(ROOT, "profiles", "r6_bench.json")))
    line = b.compact_line(r6, "bench_full.json")
    assert len(json.dumps(line)) < 4096
    assert line["config"]["plan_as_tested"] is True
    cu = line["call_us"]
    assert cu["max"] / cu["block_period_us"] <= 0.25 and cu["tail_phase_groups"] == 8          # (round 5, all channels in phase: 0.81)
    side = line["side"]
    assert set(side["host_pcie_inclusive_Msamples_s"]) == {"1024", "4096"} and side["config5_literal_roofline"]["bound"] == "fp32_fma"
    assert "bg_stream_Msamples_s" in side and set(line["other_configs"]) == {"1", "3", "5"}
    assert "rocprof_committed" in r6["roofline"] and "frac_rocprof" not in json.dumps(line)    # (full record only)
    # a multi-rank record without roofline / cpu legs still serialises
    bare = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data", "config")}
    line = b.compact_line(dict(bare, roofline=None, probe=None, cpu_baseline=None))
    assert line["roofline"] is None and line["cpu_baseline"] is None


def test_bare_gpus_n_builds_a_launcher_command(monkeypatch):
    """`python bench.py --gpus N` typed bare re-executes itself under torch.distributed.run with N ranks on 127.0.0.1 and a free
    port; with fewer devices than ranks the ranks share device 0 over gloo."""
    b = _bench()
    seen = {}

    def fake_execve(exe, cmd, env):
        seen.update(exe=exe, cmd=cmd, env=env)
        raise SystemExit(0)
    monkeypatch.setattr(b.os, "execve", fake_execve)
    monkeypatch.setattr(b.sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit):
        b.self_launch(4, 1)
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["REEVR_BENCH_SAME_DEVICE"] == "1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    with pytest.raises(SystemExit):
        b.self_launch(4, 8)
    assert "REEVR_BENCH_SAME_DEVICE" not in seen["env"] or os.environ.get("REEVR_BENCH_SAME_DEVICE") == "1"


def test_copy_crew_runs_every_chunk_exactly_once(tmp_path):
    """The staging crew of the host-pointer calls (rvc_abi.cpp CopyCrew: a few host threads that copy the channels' blocks into /
    out of the pinned staging rows; chunk claims are a compare-exchange on (job epoch, index), workers spin briefly before they
    sleep) compiled ALONE with g++ and stressed on the CPU: 30 000 jobs of 2 .. 200 chunks back to back and across worker
    sleeps -- every chunk of every job exactly once, no chunk of a finished job ever again."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    src = open(os.path.join(ROOT, "reevr_amd", "csrc", "rvc_abi.cpp")).read()
    body = src[src.index("// [copy-crew begin]"):src.index("// [copy-crew end]")]
    main = """
int main() {
  CopyCrew crew;
  std::vector<std::atomic<int>> hits(512);
  unsigned rng = 12345;
  long bad = 0;
  for (int job = 0; job < 30000; ++job) {
    rng = rng * 1664525u + 1013904223u;
    const int n = 2 + (int)((rng >> 8) % 200);
    for (int i = 0; i < 512; ++i) hits[i].store(0);
    crew.run(n, [&](int i) { hits[i].fetch_add(1); if ((i & 7) == 0) for (volatile int k = 0; k < 200; ++k) {} });
    for (int i = 0; i < 512; ++i) bad += hits[i].load() != (i < n ? 1 : 0);
    if ((job % 3000) == 0) std::this_thread::sleep_for(std::chrono::milliseconds(2));   // the workers fall asleep
  }
  std::printf("bad %ld\\n", bad);
  return bad != 0;
}
"""
    inc = "\n".join("#include <%s>" % h for h in ("algorithm", "atomic", "chrono", "condition_variable", "cstdio", "cstdlib", "functional",
                                                    "mutex", "thread", "vector"))
    cpp = tmp_path / "crew.cpp"
    cpp.write_text(inc + "\n" + body + main)
    exe = tmp_path / "crew"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", str(cpp), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], env=dict(os.environ, RVC_COPY_THREADS="4"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout + r.stderr
