#!/usr/bin/env python3
"""bench.py -- throughput of the partitioned-convolution hot path on MI355X.

Headline workload (BASELINE.json configs[1], measured the way SURVEY.md 8d defines it): stereo
instances with a 10 s IR @ 48 kHz at host block 512 (-> head 512 / tail 8192,
StereoConvolver.cpp:11-15), driven STRICTLY BLOCK-SYNCHRONOUSLY -- one process() call per
512-frame host block, exactly the plug-in's calling pattern (src/PluginProcessor.cpp:1793-1797)
-- for `--channels` lock-step channels of ONE convolver set (default 4096 = 2048 stereo instances,
each with its own IR). A single stereo pair moves 1.5 MB per block and cannot fill a 256-CU GPU;
2048 instances keep 33 GB of IR spectra + delay lines in HBM (>> the 256 MiB Infinity Cache, 11 % of
the 288 GB), which is the regime the per-block delay-line sweep (FFTConvolver.cpp:176-187) is
HBM-bound in.

Two schedules of that loop are measured in the same run, both strictly causal (nothing of a block
is used before the block has arrived) and both with the reference's partition sizes:
  * `value`: the engine's default, causal TIME TILING -- every 8th block a sweep reads a stage's IR
    spectra and delay line once and leaves partial sums for the next 8 blocks, the blocks in between
    add their few recent partitions; long delay lines get two levels of it;
  * `reference_schedule`: RVC_FLAG_NO_TIME_TILING, every block sweeps every partition like
    FFTConvolver.cpp:176-187: physical bytes = SURVEY.md 8d's algorithmic bytes, frac <= 1. Its
    fraction of the HBM peak is `roofline.alg_frac_reference_schedule`.

The set is the engine's default: thousands of block-synchronous channels are served by child sets of ~2048 channels on their
own streams (fenced internally against the set's one stream), so launches of the children -- of one kernel family and of
different ones -- share the device: `value` is that run. There a launch's duration depends on what else is running, so the
contract's per-launch `roofline` (bytes per launch / mean launch duration of the dominant kernel) is taken in the `one_queue` leg
of the same run (RVC_FLAG_NO_SUBSETS, same channels / inputs / call pattern: every launch has the device to itself), with the
default run's own figures beside it as `roofline.default_run_*` (frac = the family's bytes over the UNION of its launch
intervals, HIP events on one clock) and the whole step's executed-bytes fraction as `roofline.frac_whole_step_executed_bytes`.

Small regimes (entry `regimes`, summary in `config`): the headline loop at 2 / 16 / 64 / 256 / 1024 channels (16 channels = the
literal BASELINE config 4 on one GPU: 8 stereo instances), the literal config 5 (64 mono channels, one long call per step,
through reevr_amd.render.BatchRenderer) and the headline set with RVC_FLAG_FFT_F64 (every transform in double like the
reference's Ooura: the price of its precision at 4096 channels).

The other BASELINE configurations run in the SAME lock-step regime in the same default run
(`--configs 1,3,5`; entries `config1` / `config3` / `config5` of the line): config 1 (mono, 1 s IR,
one FFTConvolver of block 512), config 3 (30 s IR @ 96 kHz, block 256 -> head 256 / tail 8192,
350 tail partitions) and config 5's geometry (5 s IR, block 4096) -- each with both schedules and
the CPU reference timed beside it.

A "step" is `--blocks-per-step` consecutive host blocks (default 256 = 16 tail periods = 131072
frames = 2.7 s of audio per channel): 256 per-block launches + 16 tail jobs, so every step does the
same work. Inputs / outputs are resident in HBM (two batches, rotated). The timed run carries its own
correctness probe: the first and the last channel are fed unit impulses instead of noise, and their output in the LAST timed
step must be the (shifted, overlapped) impulse response.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

N > 1: instances are dealt to ranks (unit mod world), every rank runs the same
number of channels (weak scaling), no data-path collective. The gather of output blocks north_star names is ONE RCCL
all_gather per step, issued asynchronously and overlapped with the next step: configs 4 and 5 gather every channel (one
job whose outputs are collected); config 2 hosts thousands of independent instances per GPU whose outputs have no consumer
on the other GPUs, so by default it gathers the output blocks of 8 stereo instances per GPU -- config 4's size -- and
`--gather 2` gathers every channel; `--gather 0` turns it off. Under torch.distributed.run the collective also runs with
ONE rank (the RCCL code path on a 1-GPU box). One JSON line on rank 0.

`--config 4` (8 stereo instances sharded over the ranks, block-synchronous, strong scaling) and `--config 5` (64 mono
channels, 5 s IR, block 4096, offline render = one long call per step, sharded 64/N per rank, strong scaling) are the
multi-GPU forms of those configurations; `--config 1` / `--config 3` run that configuration as the headline.

Side numbers on the same line (rank 0, N = 1): one stereo pair block-synchronously (the plug-in's own case: latency per
block), the offline long-call rate of a stereo pair, and the CPU baselines.
"""
from __future__ import annotations

import argparse
import concurrent.futures
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SR = 48000
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3  # vector fp32 (same guide)
K2 = 8                    # rvc::kSweepRows: the tile the per-block patches work on
PROFILE_ROUND = "r6"      # the round whose committed rocprofv3 passes annotate the record (profiles/<round>_config<C>/, <round>_traffic.json)
TRAFFIC_JSON = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_traffic.json")

# BASELINE.json configurations as lock-step workloads: IR length, host block, single-stage?, default channels per GPU,
# host blocks per step (whole tail periods)
WORKLOADS = {
    1: dict(ir_len=48000, host_block=512, single=True, channels=8192, blocks=256,
            text="mono, 1 s IR @ 48 kHz, block=512, single FFTConvolver"),
    2: dict(ir_len=480000, host_block=512, single=False, channels=4096, blocks=256,
            text="stereo, 10 s IR @ 48 kHz, block=512 (head 512 / tail 8192), TwoStage convolver"),
    3: dict(ir_len=2880000, host_block=256, single=False, channels=2048, blocks=256,
            text="stereo, 30 s IR @ 96 kHz, block=256 (head 256 / tail 8192; 64 + 350 partitions), TwoStage convolver"),
    5: dict(ir_len=240000, host_block=4096, single=False, channels=4096, blocks=32,
            text="mono channels, 5 s IR @ 48 kHz, block=4096 (head 4096 / tail 8192), TwoStage convolver"),
}

# What the DEFAULT set of each lock-step workload runs (rvc_set_plan; everything but the channel count). bench.py checks its own
# sets against this table (`config.plan_as_tested`), and tests/test_gpu_steady_state.py compares sets with exactly these plans with
# the pinned oracle in steady state -- so the benchmark cannot drift away from what the parity tests cover.
HEADLINE_PLANS = {
    1: dict(subsets=4, two_stage=0, tail_on_second_stream=0, head_block=512, tail_block=0, zero_latency_samples=0,
            head_partitions=94, tail_partitions=0, tail_delay=0, head_f64=0, tail_f64=0, head_tile_blocks=32, tail_tile_blocks=0,
            block_path=0, head_patch_in_launch=1, head_third_level=1),
    2: dict(subsets=2, two_stage=1, tail_on_second_stream=0, head_block=512, tail_block=8192, zero_latency_samples=8192,
            head_partitions=16, tail_partitions=58, tail_delay=1, head_f64=0, tail_f64=2, head_tile_blocks=8, tail_tile_blocks=32,
            block_path=0, head_patch_in_launch=1, reference_structure=0, tail_spread=0, tail_phase_groups=8, tail_third_level=1, head_third_level=1),
    3: dict(subsets=2, two_stage=1, tail_on_second_stream=0, head_block=256, tail_block=16384, zero_latency_samples=16384,
            head_partitions=64, tail_partitions=175, tail_delay=1, head_f64=0, tail_f64=0, head_tile_blocks=32, tail_tile_blocks=32,
            block_path=0, head_patch_in_launch=1, reference_structure=0, tail_spread=0, tail_phase_groups=8, tail_third_level=1, head_third_level=0),
    5: dict(subsets=2, two_stage=1, tail_on_second_stream=0, head_block=4096, tail_block=8192, zero_latency_samples=8192,
            head_partitions=2, tail_partitions=29, tail_delay=1, head_f64=2, tail_f64=2, head_tile_blocks=0, tail_tile_blocks=16,
            block_path=1, head_patch_in_launch=0, reference_structure=0, tail_spread=0, tail_phase_groups=8, tail_third_level=1, head_third_level=0),
}


def plan_as_tested(conv, cfg: int):
    """None when the set runs HEADLINE_PLANS[cfg], else {field: (runs, tested)} -- a knob, a flag or another channel count
    changed the plan and the steady-state parity tests no longer describe this run."""
    want = HEADLINE_PLANS.get(cfg)
    if want is None:
        return {"config": (cfg, "no tested plan")}
    got = conv.plan()
    diff = {k: (got.get(k), v) for k, v in want.items() if got.get(k) != v}
    return diff or None


def alg_bytes_block(B: int, P: int) -> int:
    """algorithmic bytes of one block of one reference sub-convolver."""
    return 16 * P * (B + 1) + 8 * (B + 1) + 16 * B


def ref_partitions(head: int, tail: int, ir_len: int):
    """Partition counts of the reference's head / tail0 / tail sub-convolvers (TwoStageFFTConvolver.cpp:117-138);
    tail == 0: one FFTConvolver of block `head` (FFTConvolver.cpp:113-116)."""
    if not tail:
        return -(-ir_len // head), 0, 0
    p_head = -(-min(ir_len, tail) // head)
    p_t0 = -(-min(max(ir_len - tail, 0), tail) // head)
    p_t = -(-max(ir_len - 2 * tail, 0) // tail)
    return p_head, p_t0, p_t


def alg_bytes_per_sample(head: int, tail: int, ir_len: int) -> float:
    """Per channel-sample, reference structure head / tail0 / tail."""
    p_head, p_t0, p_t = ref_partitions(head, tail, ir_len)
    if not tail:
        return alg_bytes_block(head, p_head) / head
    per_tail_block = (tail // head) * (alg_bytes_block(head, p_head) + (alg_bytes_block(head, p_t0) if p_t0 else 0))
    per_tail_block += alg_bytes_block(tail, p_t) if p_t else 0
    return per_tail_block / tail


def flops_per_sample(head: int, tail: int, ir_len: int) -> float:
    """SURVEY.md 8d flop model: 2 * 2.5 N log2 N + 8 P (B+1) per block of each sub-convolver."""
    def blk(B, P):
        N = 2 * B
        return 2 * 2.5 * N * np.log2(N) + 8 * P * (B + 1)
    p_head, p_t0, p_t = ref_partitions(head, tail, ir_len)
    if not tail:
        return float(blk(head, p_head) / head)
    per = (tail // head) * (blk(head, p_head) + (blk(head, p_t0) if p_t0 else 0)) + (blk(tail, p_t) if p_t else 0)
    return float(per / tail)


def make_irs(ir_len: int, instances, distinct: int = 0):
    """One synthetic stereo IR per instance (reevr_amd.synth), generated in parallel. distinct > 0: only
    that many different instances are synthesised and cycled through (every channel still gets its OWN spectra in HBM)."""
    from reevr_amd import synth
    ids = list(instances)
    uniq = sorted(set(i % distinct for i in ids)) if distinct else sorted(set(ids))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        sets = dict(zip(uniq, ex.map(lambda i: synth.synth_ir(ir_len, 2, inst=i), uniq)))
    return [sets[i % distinct if distinct else i][c] for i in ids for c in range(2)]


def cpu_baseline(irs, x, head_block: int, tail: int, budget_s: float, what: str) -> dict:
    """The reference itself (oracle/_ref, kind "reference") or the C restatement (kind "port") on this
    host's cores: host-block-sized process() calls back to back, tail inline, one instance per thread, driven
    by a pthread loop in C (oracle/cpu_bench.c -- no Python in the loop). Bounded to ~budget_s."""
    from oracle import oracle_py as O
    which = "ref" if O.have_ref() else "orc"
    try:
        cores = len(os.sched_getaffinity(0))          # the CPUs this process may run on (a container's share of the host)
    except AttributeError:
        cores = os.cpu_count() or 1
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    n1, w1, _ = O.cpu_bench(which, 1, head_block, tail, head_block, irs[:1], x[:1], budget_s * 0.4)
    nm, wm, per = O.cpu_bench(which, cores, head_block, tail, head_block, irs, x, budget_s * 0.6)
    return {
        "value": round(n1 / w1 / 1e6, 3), "unit": "Msamples/s", "cores": 1,
        "kind": "reference" if which == "ref" else "port",
        "sample": f"mono instance, {what}, {head_block}-frame process() calls back to back, "
                  f"tail inline: {n1} samples in {w1:.1f} s on 1 thread (C loop, oracle/cpu_bench.c)",
        "cpu_model": model, "flags": "g++ -O3 (SSE2 MAC as shipped, Utilities.cpp:62-111)",
        "all_cores": {"value": round(nm / wm / 1e6, 3), "cores": cores,
                      "median_thread_Msamples_s": round(float(np.median(per)) / wm / 1e6, 3),
                      "sample": f"{cores} independent mono instances ({len(irs)} distinct IRs), one pthread each, "
                                f"{nm} samples in {wm:.1f} s; slowest / fastest thread {min(per)} / {max(per)} samples "
                                "(a fixed wall budget on a shared host: the spread between threads and between runs is the "
                                "host's scheduling, quote the 1-thread figure)"},
    }


def geometry(host_block: int, single: bool):
    head = 1
    while head < host_block:
        head *= 2
    return head, (0 if single else max(8192, 2 * head))                # StereoConvolver.cpp:11-15


def executed_bytes(conv, nch: int, head: int, tail: int, ir_len: int, host_block: int, tiled: bool) -> dict:
    """Bytes per LAUNCH of each kernel family. Reference schedule: SURVEY.md 8d's algorithmic figures (what the
    reference's loop nest moves = what these kernels move). Time-tiled schedule: the bytes of the structure
    actually executed, so that frac <= 1 means what it says."""
    n1 = nch // max(1, conv.subsets)                       # channels per launch (child sets launch separately)
    PA, PT = conv.partitions(0), conv.partitions(1)
    p_head, p_t0, p_t = ref_partitions(head, tail, ir_len)
    # (the tail block the set RUNS: long tails of many-channel sets are served at twice the requested block with delay 1)
    tail_x = int(conv.tail_block) if tail else 0
    row_h, row_t = 8.0 * head * n1, 8.0 * max(tail_x, 1) * n1   # bytes of one spectrum row of every channel of a launch
    io_blk = n1 * (4.0 * (3 if tail else 2) * head + 4.0 * 2 * head)   # per block: input + history (+ tail ring) read, output + ring written
    if not tiled:
        exe = {
            # the per-block launch does one head block of the reference's head AND tail0 sub-convolvers
            "fused_block": float(n1 * (alg_bytes_block(head, p_head) + (alg_bytes_block(head, p_t0) if p_t0 else 0))),
            "premultiply": float(n1 * 16 * PA * (head + 1)),
            "fir_head": float(n1 * 16 * PA * (head + 1)),
            "fft_fwd_head": float(n1 * (4 * head + 8 * (head + 1))),
            "fft_inv_head": float(n1 * (8 * (head + 1) + 12 * head)),
            "ingest": float(n1 * 8 * host_block),
        }
        if tail:
            exe.update({"fir_tail": float(n1 * 16 * p_t * (tail + 1)),
                        "fft_fwd_tail": float(n1 * (4 * tail + 8 * (tail + 1))),
                        "fft_inv_tail": float(n1 * (8 * (tail + 1) + 12 * tail))})
        return exe
    KA, KT = conv.tile_rows(0), conv.tile_rows(1)

    def sweep2_rows(K1, P, delay):       # second-level sweep: mean over the groups 1 .. K1/8 - 1 of a first-level tile
        gs = range(1, K1 // K2)
        return float(np.mean([min(P, K2 * g + K2 + 1 - delay) + K2 * g + 2 * K2 for g in gs])) if K1 > K2 else 0.0
    exe = {}
    if KA:
        if conv.plan()["head_patch_in_launch"]:
            # round 5: the launch patches its OWN block and hands the row over through LDS -- audio part (H0, H1, X_{k-1} read; X_k
            # written; samples) + the block's sweep row + on average (K2-1)/2 recent partitions (an IR row and a delay-line row each)
            # round 6: the stage's sweeps take the newest row too (a patch is one partition shorter), and with head_third_level a
            # third-level sweep half way through every group of 8 lets the last four blocks' patches start over
            third_h = bool(conv.plan().get("head_third_level", 0))
            mean_p = float(np.mean([max(0, (j % (K2 // 2) if third_h else j) - 1) for j in range(K2)]))
            exe["fused_block"] = 4 * row_h + io_blk + (2 * mean_p + 1) * row_h
            if third_h:          # 6 IR rows, 4 delay-line rows, the group's 4 rows read, 4 written
                exe["sweep3_head"] = 18.0 * row_h
        else:
            # audio part (H0, H1, X_{k-1}, accumulator read; X_k written; samples) + on average (K2-1)/2 recent partitions patched
            exe["fused_block"] = 5 * row_h + io_blk + ((K2 - 1) / 2.0 * 2 + 2) * row_h * (K2 - 1) / K2
        # IR rows 2.. + arrived delay-line rows read once, KA partial rows written (the two newest partitions are the per-block launch's)
        # (same-block sets: the sweep takes the newest row too -- one more delay-line row)
        exe["sweep_head"] = ((PA - 2) + (PA - 2) + (1 if conv.plan()["head_patch_in_launch"] else 0) + KA) * row_h
        # (same-block sets: the stage's sweeps take the newest row -- lag 1 -- so a second-level walk is one partition shorter)
        exe["sweep2_head"] = sweep2_rows(KA, PA - 2, 3 if conv.plan()["head_patch_in_launch"] else 2) * row_h
    else:                                                    # zero-latency stage not tiled: every block reads all of it
        exe["fused_block"] = 5 * row_h + io_blk + (2.0 * max(PA - 2, 0) + 1) * row_h
        # (many channels with a large head block: the per-block call is transform / delay line / inverse launches)
        # (the head transform reads the block from the caller's buffer and appends it to the ring itself: + 4 * head, no ingest launch)
        exe.update({"fft_fwd_head": float(n1 * (4 * 2 * head + 8 * head + 4 * head)),
                    "fir_head": (2.0 * PA + 1) * row_h,
                    # (spectrum row in, the tail stage's stream in, samples out)
                    "fft_inv_head": float(n1 * (8 * head + (4 * head if tail and PT else 0) + 4 * head))})
    exe["premultiply"] = 2.0 * max(PA - 2, 0) * row_h + row_h
    if tail and PT:
        if KT:
            # (spread sweeps, rvc_plan::tail_spread: a sweep is `slices` launches of a share of the channels each, leaves its
            #  newest row to the patches -- one partition more per patch -- and the second-level window is one row longer)
            pl = conv.plan()
            sp, slices = pl.get("tail_spread", 0), max(1, pl.get("tail_sweep_slices", 1))
            G = max(1, pl.get("tail_phase_groups", 1))   # phase groups: every sweep / patch launch covers 1 / G of the channels
            l1, l2 = sp & 1, (sp >> 1) & 1
            exe["sweep_tail"] = (2 * PT + KT) * row_t / (slices if l1 else 1) / G
            exe["sweep2_tail"] = (sweep2_rows(KT, PT, 2) + 2 * l1) * row_t / (slices if l2 else 1) / G
            # patch: the sweep row + the recent partitions (a group's first block needs none when its sweep is not spread), 1 row out
            # third level (rvc_plan::tail_third_level): half way through a group a sweep over the 4 rows that arrived since (7 IR rows, 4
            # delay-line rows, the group's 4 rows read, 4 written) gives the last four blocks rows of their own: their patches start over
            third = bool(pl.get("tail_third_level", 0))
            if third:
                exe["sweep3_tail"] = 19.0 * row_t / G
            depth = [(j - K2 // 2) if (third and j >= K2 // 2) else j + (l1 if g == 0 else l2) for g in range(max(1, KT // K2)) for j in range(K2)]
            per = [2 * d + 2 for d in depth if d > 0]
            exe["fir_tail"] = float(np.mean(per)) * row_t
            if G > 1:      # phase groups: ONE patch launch per tail block over all channels, every group at its own depth (depth 0: the
                # group's sweep of that block wrote the row in place, FirArgs::Y0 -- nothing to do)
                exe["fir_tail"] = float(np.mean([2 * d + 2 if d > 0 else 0 for d in depth])) * row_t
        else:
            exe["fir_tail"] = (2.0 * PT + 1) * row_t
        exe["fft_fwd_tail"] = float(n1 * (4 * 2 * tail_x + 8 * tail_x))
        exe["fft_inv_tail"] = float(n1 * (8 * tail_x + 4 * tail_x))
    return exe


def tail_stage_form(conv, head: int, tail: int) -> str:
    """What the set's tail stage looks like beside the reference's (head over IR[0,T), tail0 over IR[T,2T), tail over IR[2T,..) at
    block T, two blocks late): the engine's zero-latency stage covers head + tail0; lock-step sets of many channels run the tail
    with delay ONE and spend the freed period on a tail at block 2T (long tails) or on half the zero-latency stage."""
    if not tail or not conv.partitions(1):
        return "none"
    if int(conv.tail_block) != tail:
        return "delay 1, block 2T over IR[2T,..) (widened)"
    if conv.partitions(0) * head <= tail:
        return "delay 1, block T over IR[T,..) (zero-latency stage shrunk to IR[0,T))"
    return "delay 2, block T over IR[2T,..) (the reference's structure)"


def transforms_form(conv) -> str:
    """Which transforms of the set run in double (rvc_set_plan head_f64 / tail_f64: bit 0 forward, bit 1 inverse)."""
    p = conv.plan()
    names = {0: "f32", 1: "forward f64", 2: "inverse f64", 3: "f64"}
    return "zero-latency stage %s, tail stage %s" % (names[p["head_f64"]], names[p["tail_f64"]] if p["tail_partitions"] else "-")


def probe_expected(ir: np.ndarray, frames_step: int, last_step: int, at: int) -> np.ndarray:
    """Channel 0 is fed a unit impulse at offset `at` of input batch 0, i.e. at absolute sample s * frames_step + at of
    every EVEN step s: its output in step `last_step` is the sum of the impulse responses those impulses started."""
    want = np.zeros(frames_step, np.float64)
    for s in range(0, last_step + 1, 2):
        off = (last_step - s) * frames_step - at             # IR index of the step's first sample
        lo = max(0, -off)
        n = min(frames_step, len(ir) - off)
        if n > lo:
            want[lo:n] += ir[off + lo:off + n]
    return want


class Lockstep:
    """One lock-step workload on this rank: the set, its resident input / output batches, the step function."""

    def __init__(self, torch, reevr_amd, synth, cfg: int, instances, local_rank: int, tiling: bool, bg: bool,
                 blocks_per_step: int, long_call: bool = False, distinct: int = 0, irs=None, x=None, child_sets=None,
                 fft_f64: bool = False, fft_f64_long: bool = False, fft_f32: bool = False):
        self.torch, self.cfg = torch, cfg
        w = WORKLOADS[cfg if cfg in WORKLOADS else 2]
        self.ir_len, self.host_block, self.single = w["ir_len"], w["host_block"], w["single"]
        self.long_call = long_call
        self.head, self.tail = geometry(self.host_block, self.single)
        self.nch = 2 * len(instances)
        self.frames_step = 20 * SR if long_call else blocks_per_step * self.host_block
        self.frames_step -= self.frames_step % self.host_block
        self.nbuf = 1 if long_call else 2
        self.dev = torch.device("cuda", local_rank)
        self.probe_at = 5
        self.last_out = None
        t0 = time.perf_counter()
        self.irs = irs if irs is not None else make_irs(self.ir_len, instances, distinct)
        if x is None:
            with concurrent.futures.ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
                x = np.stack(list(ex.map(lambda uc: synth.synth_input(self.frames_step * self.nbuf, 2 * uc[0] + uc[1]),
                                         [(u, c) for u in instances for c in range(2)])))
            if not long_call:                  # correctness probe: channel 0 gets unit impulses instead of noise
                for pc in {0, x.shape[0] - 1}:      # (first and last channel: with child sets, one in the first and one in the last child)
                    x[pc, :] = 0.0
                    x[pc, self.probe_at] = 1.0
        self.x = x
        self.synth_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        max_len = self.frames_step if long_call else self.host_block
        self.renderer = None
        if long_call:      # BASELINE config 5 as written: the raw batch renderer (reevr_amd/render.py)
            from reevr_amd.render import BatchRenderer
            self.renderer = BatchRenderer(self.irs, self.host_block, max_len, device=local_rank)
            self.conv = self.renderer.set
            assert (self.renderer.head, self.renderer.tail) == (self.head, self.tail)
        else:
            self.conv = reevr_amd.ConvolverSet(self.nch, device=local_rank, bg_stream=bg, time_tiling=tiling, child_sets=child_sets,
                                               fft_f64=fft_f64, fft_f64_long=fft_f64_long, fft_f32=fft_f32)
            ok = (self.conv.init_uniform(self.host_block, self.irs, max_len=max_len) if self.single
                  else self.conv.init(self.host_block, self.tail, self.irs, max_len=max_len))
            if not ok:
                raise SystemExit(f"init failed: {self.conv.last_error_string}")
        self.conv.sync()
        # the tail block the set runs (the engine serves long tails of many-channel sets at twice the requested block, delay 1)
        self.tail_used = int(self.conv.tail_block) if self.tail else 0
        self.init_ms = (time.perf_counter() - t0) * 1e3
        self.d_in = torch.from_numpy(self.x).to(self.dev)
        self.d_out = torch.empty_like(self.d_in)
        torch.cuda.synchronize()
        self.i = 0
        self.tiled = bool(tiling) and (self.conv.tile_rows(0) > 0 or self.conv.tile_rows(1) > 0)

    def batch(self, b):
        f = self.frames_step
        return self.d_in[:, b * f:(b + 1) * f], self.d_out[:, b * f:(b + 1) * f]

    def step(self, out=None, order=False):
        b = self.i % self.nbuf
        self.i += 1
        xi, yo = self.batch(b)
        if out is not None:
            yo = out
        if self.long_call:
            self.renderer.process_device(xi, yo, sync=False, order=order)
        else:                                  # the host's per-block loop (in C): one call per host block
            self.conv.process_device_blocks(xi, self.host_block, yo, sync=False, order=order)
        self.last_out = yo
        return b, yo

    def preroll(self):
        """untimed run to the steady state: the tail delay line holds P_T tail blocks of history and rows before
        time 0 are never fetched, so the first P_T + 2 tail periods move fewer bytes"""
        pre = 0
        if not self.long_call:
            span = self.tail_used if self.tail else self.head
            parts = self.conv.partitions(1) if self.tail else self.conv.partitions(0)
            pre = -(-(parts + 4) * span // self.frames_step)
            for _ in range(pre):
                self.step()
            self.conv.sync()
        else:
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < 0.05:
                self.step()
                self.conv.sync()
        return pre

    def check_probe(self):
        """the last step's output of the probe channels (the first and the last: unit impulses in, see probe_expected) against the
        impulse responses their impulses started; reported: the worse of the two"""
        if self.long_call or self.i == 0:
            return None
        last = self.i - 1
        worst = None
        for pc in sorted({0, self.nch - 1}):
            if not np.array_equal(self.x[pc, :self.probe_at + 1], np.eye(1, self.probe_at + 1, self.probe_at, dtype=self.x.dtype)[0]):
                continue                      # (inputs handed over from another leg without a probe on this channel)
            got = self.last_out[pc].cpu().numpy().astype(np.float64)
            want = probe_expected(self.irs[pc].astype(np.float64), self.frames_step, last, self.probe_at)
            err = float(np.sqrt(np.mean((got - want) ** 2)))
            ref = float(np.sqrt(np.mean(want ** 2)))
            rec = {"channel": pc, "step": last, "rms_error": err, "rms_expected": ref,
                   "ok": bool(err <= 1e-5 * max(ref, 1e-12) + 1e-9)}
            if worst is None or not rec["ok"] or (worst["ok"] and err > worst["rms_error"]):
                worst = dict(rec, channels_checked=sorted({0, self.nch - 1}))
        return worst

    def kernel_times(self, KERNEL_NAMES):
        """per-kernel durations, live, with HIP events on the streams the kernels run on -- over as many steps as one
        first-level sweep tile of the tail stage spans (so that every kernel family occurs), reported per step"""
        span = self.tail_used if self.tail else self.head
        k1 = self.conv.tile_rows(1 if self.tail else 0) or 1
        nsteps = 1 if self.long_call else max(1, -(-k1 * span // self.frames_step))
        self.conv.set_timing(True)
        self.conv.kernel_time_reset()
        for _ in range(nsteps):
            self.step()
        self.conv.sync()
        kern = {}
        for kid, name in enumerate(KERNEL_NAMES):
            n, ms = self.conv.kernel_time(kid)
            if n:
                # launches of child sets overlap in time: the union of the family's intervals is the time it kept the
                # device busy (= the sum of the durations when the set has no children)
                iv = self.conv.kernel_intervals(kid)
                busy, end = 0.0, -1e30
                for a, b in iv[np.argsort(iv[:, 0])]:
                    if b > end:
                        busy += b - max(a, end)
                        end = b
                kern[name] = {"launches_per_step": float(n) / nsteps, "avg_ms": ms / n,
                              "busy_ms_per_step": (busy if len(iv) == n else ms) / nsteps}
        self.conv.set_timing(False)
        self.conv.kernel_time_reset()
        return kern

    def call_latency(self, steps: int = 0) -> dict:
        """What each per-block call costs the device in the back-to-back loop: the block loop with a completion stamp behind
        every call (HIP events on the set's streams, every child set; rvc_set_process_device_blocks_stamped), over whole
        first-level tiles. The reference evens this out with its background thread (src/dsp/Convolver.cpp:84-95); here the
        tail stage's sweeps are spread over the calls of a tail period (rvc_plan::tail_spread)."""
        if self.long_call:
            return {}
        steps = steps or max(2, self.tile_period_steps())
        gaps = []
        for _ in range(steps):
            b = self.i % self.nbuf
            self.i += 1
            xi, yo = self.batch(b)
            _, done = self.conv.process_device_blocks_stamped(xi, self.host_block, yo)
            self.last_out = yo
            gaps.append(np.diff(done))                       # (the step's first call carries the loop's start-up: dropped)
        g = np.sort(np.concatenate(gaps)) * 1e3
        sr = 96000 if self.cfg == 3 else SR
        period_us = self.host_block / sr * 1e6
        p = self.conv.plan()
        return {"calls": int(len(g)), "mean": round(float(g.mean()), 2), "p50": round(float(g[len(g) // 2]), 2),
                "p99": round(float(g[int(len(g) * 0.99)]), 2), "max": round(float(g[-1]), 2), "unit": "us",
                "block_period_us": round(period_us, 1), "max_over_block_period": round(float(g[-1]) / period_us, 4),
                "tail_spread": p.get("tail_spread", 0), "sweep_slices": p.get("tail_sweep_slices", 1),
                "tail_phase_groups": p.get("tail_phase_groups", 1)}

    def tile_period_steps(self) -> int:
        """steps one first-level tile of the longest-tiled stage spans: a first-level sweep runs once per tile, so only a whole
        number of tiles is a fair average (config 3: 32 tail blocks = 4 steps of 8; configs 1 / 2 / 5: 1)"""
        if self.long_call:
            return 1
        span = self.tail_used if self.tail else self.head
        k1 = self.conv.tile_rows(1 if self.tail else 0) or 1
        return max(1, -(-k1 * span // self.frames_step))

    def timed(self, steps: int, warmup: int, whole_tiles: bool = False):
        if whole_tiles:                        # (side measurements: round the step count up to whole first-level tiles)
            p = self.tile_period_steps()
            steps = -(-steps // p) * p
        self.timed_steps = steps
        for _ in range(warmup):
            self.step()
        self.conv.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.conv.sync()
        el = time.perf_counter() - t0
        return self.nch * self.frames_step * steps / el, el / steps * 1e3

    def close(self):
        self.conv.close()
        del self.d_in, self.d_out
        self.torch.cuda.empty_cache()


def roofline_tables(kern: dict, exe: dict, traffic: dict):
    """Per kernel family: bytes of the executed structure over the time the family kept the device busy. A set without
    children: busy time = launches x average duration (the contract's per-launch figure). With child sets two launches of
    a family share the device: per-launch durations (`avg_launch_ms`, what a profiler lists) count that shared time
    twice, so `frac` is taken over the UNION of the family's launch intervals and `frac_per_launch` is the literal
    bytes-per-launch / average-duration figure; `concurrency` = sum of durations / union."""
    roof_all, exe_bytes_step = {}, 0.0
    for k, v in kern.items():
        if k not in exe or exe[k] <= 0:
            continue
        tot = exe[k] * v["launches_per_step"]
        busy = v.get("busy_ms_per_step", v["avg_ms"] * v["launches_per_step"])
        gbs = tot / (busy * 1e-3) / 1e9
        lit = exe[k] / (v["avg_ms"] * 1e-3) / 1e9
        exe_bytes_step += tot
        roof_all[k] = {"launches_per_step": v["launches_per_step"], "avg_launch_ms": round(v["avg_ms"], 5),
                       "ms_per_step": round(busy, 5), "concurrency": round(v["avg_ms"] * v["launches_per_step"] / busy, 3),
                       "bytes_per_launch": round(exe[k]), "achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                       "frac_per_launch": round(lit / HBM_PEAK_GBS, 4), "traffic": round(traffic[k]) if k in traffic else None}
    return roof_all, exe_bytes_step


def load_traffic(nch_per_launch: int, cfg: int, tiled: bool):
    """Counter-measured HBM bytes per launch of each kernel family (profiles/<PROFILE_ROUND>_traffic.json: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes of this command, gfx950-corrected). An entry only applies to launches of exactly the
    channel count it was measured with (`channels_per_launch`: the set's channels / its child sets): anything else is
    refused -- a figure taken at another launch size reads like a model error."""
    if not os.path.exists(TRAFFIC_JSON):
        return {}, None
    tj = json.load(open(TRAFFIC_JSON))
    ents = [e for e in tj.values() if isinstance(e, dict) and e.get("config") == cfg and
            e.get("channels_per_launch") == nch_per_launch and bool(e.get("time_tiling", 0)) == tiled]
    if not ents:
        return {}, None
    ent = ents[0]
    return ({k: v["traffic_bytes"] for k, v in ent.get("kernels", {}).items()},
            "profiles/" + PROFILE_ROUND + "_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950-corrected; "
            "%d channels per launch)" % nch_per_launch)


def side_config(torch, reevr_amd, synth, KERNEL_NAMES, cfg: int, channels: int, local_rank: int, steps: int, cpu_s: float):
    """One of the other BASELINE configurations in the lock-step regime (single GPU): the engine's schedule, the
    reference's schedule (RVC_FLAG_NO_TIME_TILING) on the same inputs, and the CPU reference beside them."""
    w = WORKLOADS[cfg]
    inst = list(range(channels // 2))
    ls = Lockstep(torch, reevr_amd, synth, cfg, inst, local_rank, True, False, w["blocks"], distinct=128)
    pre = ls.preroll()
    rate, ms = ls.timed(steps, 2, whole_tiles=True)
    steps = ls.timed_steps
    probe = ls.check_probe()
    ls.conv.check()
    kern = ls.kernel_times(KERNEL_NAMES)
    exe = executed_bytes(ls.conv, ls.nch, ls.head, ls.tail, ls.ir_len, ls.host_block, ls.tiled)
    traffic, tsrc = load_traffic(ls.nch // max(1, ls.conv.subsets), cfg, ls.tiled)
    roof_all, exe_step = roofline_tables(kern, exe, traffic)
    bps = alg_bytes_per_sample(ls.head, ls.tail, ls.ir_len)
    exe_bps = exe_step / (ls.nch * ls.frames_step) if exe_step else None
    out = {
        "workload": f"{w['text']}: {channels} lock-step channels, one process() per {ls.host_block}-frame block",
        "value": round(rate / 1e6, 3), "unit": "Msamples/s", "steps": steps, "ms_per_step": round(ms, 4),
        "channels": channels, "frames_per_channel_per_step": ls.frames_step, "pre_roll_steps": pre,
        "partitions": {"zero-latency stage": ls.conv.partitions(0), "tail stage": ls.conv.partitions(1)},
        "tail_block_run": ls.tail_used, "tail_stage": tail_stage_form(ls.conv, ls.head, ls.tail),
        "tile_blocks": {"zero-latency stage": ls.conv.tile_rows(0), "tail stage": ls.conv.tile_rows(1)},
        "subsets": ls.conv.subsets, "plan_as_tested": plan_as_tested(ls.conv, cfg) is None,
        "executed_bytes_per_sample": round(exe_bps, 1) if exe_bps else None,
        "frac_of_hbm_peak_executed_bytes": round(rate * exe_bps / 1e9 / HBM_PEAK_GBS, 4) if exe_bps else None,
        "alg_bytes_per_sample": round(bps, 1),
        "alg_equiv": round(rate * bps / 1e9 / HBM_PEAK_GBS, 4),
        "probe": probe, "roofline_all": roof_all, "traffic_source": tsrc,
        "init_ms": round(ls.init_ms, 1), "synth_s": round(ls.synth_s, 1),
    }
    irs, x = ls.irs, ls.x
    ls.close()
    # the same loop in the reference's sweep order (same channels, same inputs)
    rs = Lockstep(torch, reevr_amd, synth, cfg, inst, local_rank, False, False, w["blocks"], irs=irs, x=x)
    rs.preroll()
    rsteps = max(2, steps // 4)
    rrate, rms = rs.timed(rsteps, 1)
    rs.conv.check()
    rs.close()
    out["reference_schedule"] = {"value": round(rrate / 1e6, 3), "unit": "Msamples/s", "steps": rsteps,
                                 "ms_per_step": round(rms, 4), "achieved_GBs": round(rrate * bps / 1e9, 1),
                                 "alg_frac": round(rrate * bps / 1e9 / HBM_PEAK_GBS, 4)}
    out["alg_frac_reference_schedule"] = out["reference_schedule"]["alg_frac"]
    # ... and on ONE queue (RVC_FLAG_NO_SUBSETS): every launch has the device to itself, bytes per launch / mean launch duration
    # is then an efficiency per family (what a profiler's per-kernel average corresponds to)
    if out["subsets"] > 1:
        oq = Lockstep(torch, reevr_amd, synth, cfg, inst, local_rank, True, False, w["blocks"], irs=irs, x=x, child_sets=False)
        oq.preroll()
        osteps = max(2, steps // 2)
        orate, oms = oq.timed(osteps, 1, whole_tiles=True)
        osteps = oq.timed_steps
        oprobe = oq.check_probe()
        oq.conv.check()
        okern = oq.kernel_times(KERNEL_NAMES)
        oexe = executed_bytes(oq.conv, oq.nch, oq.head, oq.tail, oq.ir_len, oq.host_block, oq.tiled)
        otraffic, _ = load_traffic(oq.nch, cfg, oq.tiled)
        oroof, _ = roofline_tables(okern, oexe, otraffic)
        out["one_queue"] = {"value": round(orate / 1e6, 3), "unit": "Msamples/s", "steps": osteps, "ms_per_step": round(oms, 4),
                            "probe_ok": bool(oprobe and oprobe["ok"]),
                            "roofline_all": {k: {"launches_per_step": v["launches_per_step"], "avg_launch_ms": v["avg_launch_ms"],
                                                 "bytes_per_launch": v["bytes_per_launch"], "frac": v["frac"], "traffic": v["traffic"]}
                                             for k, v in oroof.items()}}
        oq.close()
    if cpu_s > 0:
        cores = os.cpu_count() or 1
        n_ir = min(len(irs), max(2, cores))
        xin = [np.ascontiguousarray(x[1 + c % (len(x) - 1)]) for c in range(n_ir)]      # (channel 0 carries the probe)
        cb = cpu_baseline(irs[:n_ir], xin, ls.host_block, ls.tail, cpu_s, w["text"])
        # (compact: the headline's cpu_baseline carries the host's description once)
        out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": 1, "kind": cb["kind"],
                               "all_cores": {"value": cb["all_cores"]["value"], "cores": cb["all_cores"]["cores"]}}
    return out


def literal_roofline(ls, rate: float, kern: dict) -> dict:
    """Roofline of the literal config 5 leg (one long call per step through the whole-IR delay line at the largest block the
    call allows -- rvc_plan::wide_block / long_call_block). Two readings: SURVEY.md 8d's flop model of the REFERENCE's structure
    (656 flop per sample) x the measured rate against the fp32 vector peak -- a throughput equivalent like `alg_equiv`: the long
    call runs fewer, larger partitions than the reference --, and per kernel family what the launch executes: the delay-line
    kernel's multiply-adds (8 flop per partition x bin x output row) and the transforms' bytes over their live durations."""
    p = ls.conv.plan()
    nch, frames = ls.nch, ls.frames_step
    B = int(p["wide_block"] or p["long_call_block"] or p["tail_block"] or p["head_block"])
    P = ls.conv.partitions(2) if p["wide_block"] else (ls.conv.partitions(1) + 2 if p["long_call_block"] else ls.conv.partitions(0))
    rows = -(-frames // B) + 1                                     # output rows of one call (the partly filled last block included)
    fps = flops_per_sample(ls.head, ls.tail, ls.ir_len)
    out = {"bound": "fp32_fma", "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
           "achieved": round(rate * fps / 1e12, 2), "frac": round(rate * fps / 1e12 / FP32_PEAK_TFLOPS, 4),
           "flops_per_sample_model": round(fps, 1), "block": B, "partitions": P, "rows_per_call": rows, "families": {}}
    fam_bytes = {"fft_fwd_tail": nch * rows * (4.0 * 2 * B + 8.0 * B), "fft_inv_tail": nch * rows * (8.0 * B + 4.0 * B),
                 # the LDS-tiled delay line: every X row once per bin tile walk, the IR rows once per 64-row time tile, Y rows out
                 "fir_tail": nch * 8.0 * B * (rows + P * -(-rows // 64) + rows)}
    for k, v in kern.items():
        ent = {"ms_per_call": round(v["avg_ms"] * v["launches_per_step"], 4)}
        if k in fam_bytes:
            ent["GBs"] = round(fam_bytes[k] / (v["avg_ms"] * v["launches_per_step"] * 1e-3) / 1e9, 1)
            ent["frac_hbm"] = round(ent["GBs"] / HBM_PEAK_GBS, 4)
        if k == "fir_tail":
            tf = 8.0 * P * B * rows * nch / (v["avg_ms"] * v["launches_per_step"] * 1e-3) / 1e12
            ent["executed_TFLOPs"] = round(tf, 2)
            ent["frac_fp32"] = round(tf / FP32_PEAK_TFLOPS, 4)
        out["families"][k] = ent
    return out


def small_regimes(torch, reevr_amd, synth, KERNEL_NAMES, local_rank: int, steps: int, irs4096=None, x4096=None, instances=None):
    """The regimes between one stereo pair and the headline's thousands of channels (where BASELINE configs 4 and 5 live on
    an 8-GPU node), each with the impulse probe; plus the literal config 5 and the headline set with double transforms."""
    out = {"channel_sweep": {}}
    for ch in (2, 16, 64, 256, 1024):
        ls = Lockstep(torch, reevr_amd, synth, 2, list(range(ch // 2)), local_rank, True, False, WORKLOADS[2]["blocks"], distinct=min(64, ch // 2))
        pre = ls.preroll()
        rate, ms = ls.timed(steps, 1)
        probe = ls.check_probe()
        ls.conv.check()
        kern = ls.kernel_times(KERNEL_NAMES)
        exe = executed_bytes(ls.conv, ls.nch, ls.head, ls.tail, ls.ir_len, ls.host_block, ls.tiled)
        _, exe_step = roofline_tables(kern, exe, {})
        exe_bps = exe_step / (ls.nch * ls.frames_step) if exe_step else None
        out["channel_sweep"][str(ch)] = {
            "value": round(rate / 1e6, 2), "unit": "Msamples/s", "us_per_block": round(ms * 1e3 / (ls.frames_step // ls.host_block), 3),
            "frac_of_hbm_peak_executed_bytes": round(rate * exe_bps / 1e9 / HBM_PEAK_GBS, 4) if exe_bps else None,
            "time_tiled": ls.tiled, "subsets": ls.conv.subsets, "probe_ok": bool(probe and probe["ok"]), "pre_roll_steps": pre}
        ls.close()
    out["channel_sweep"]["note"] = ("the headline loop (config 2's geometry, one process_device() per 512-frame block) at other channel counts; "
                                    "16 channels = BASELINE config 4 on ONE GPU (8 stereo instances); on the 8-GPU node configs 4 / 5 put 2 / 8 "
                                    "channels on each GPU")
    # BASELINE config 5 as written: 64 parallel mono channels, 5 s IR, block 4096, one long call per step (the raw batch renderer)
    l5 = Lockstep(torch, reevr_amd, synth, 5, list(range(32)), local_rank, True, False, 0, long_call=True)
    l5.preroll()
    rate5, ms5 = l5.timed(max(steps, 10), 2)
    l5.conv.check()
    out["config5_literal"] = {"value": round(rate5 / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(ms5, 4), "channels": 64,
                              "frames_per_channel_per_step": l5.frames_step,
                              "workload": "64 mono channels, 5 s IR @ 48 kHz, block 4096, ONE process() of 20 s per step through "
                                          "reevr_amd.render.BatchRenderer (the raw many-channel renderer, SURVEY 8f f-4)",
                              "roofline": literal_roofline(l5, rate5, l5.kernel_times(KERNEL_NAMES))}
    l5.close()
    # the headline set with other transform precisions: float throughout (RVC_FLAG_FFT_F32: the default of rounds 1-4; the default
    # since round 5 runs the tail stage's INVERSE transform in double, which is what takes the reference's own known-answer rule,
    # Test.cpp:129-145, on board for lock-step sets), both tail transforms in double (RVC_FLAG_FFT_F64_LONG), everything in double
    if instances is not None:
        for key, kw, nsteps, note in (
                ("fft_f32", dict(fft_f32=True), 6, "RVC_FLAG_FFT_F32: float transforms throughout (the reference's rule then fails by up to 10 % "
                                                    "on 4 of its 58 cases; 1e-5 RMS holds 100x over)"),
                ("fft_f64_long", dict(fft_f64_long=True), 4, "RVC_FLAG_FFT_F64_LONG: both 8192-bin tail transforms in double"),
                ("fft_f64", dict(fft_f64=True), 2, "RVC_FLAG_FFT_F64: every transform in double (Ooura's precision, AudioFFT.cpp:114-159); the "
                                                   "per-block call then takes the general path (the one-launch block kernel is float only)")):
            lf = Lockstep(torch, reevr_amd, synth, 2, instances, local_rank, True, False, WORKLOADS[2]["blocks"], irs=irs4096, x=x4096, **kw)
            lf.preroll()
            ratef, msf = lf.timed(nsteps, 1)
            pf = lf.check_probe()
            lf.conv.check()
            out[key] = {"value": round(ratef / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(msf, 4), "channels": lf.nch,
                        "subsets": lf.conv.subsets, "probe_ok": bool(pf and pf["ok"]), "probe_rms_error": pf["rms_error"] if pf else None,
                        "note": note}
            lf.close()
    return out


def self_launch(n: int, n_devices: int):
    """`python bench.py --gpus N` typed bare (no WORLD_SIZE in the environment): start the N ranks ourselves -- this process
    becomes `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same arguments>` on a free port, one rank
    per GPU over RCCL; rank 0 prints the line. On a box with fewer than N devices the ranks share device 0 over gloo
    (REEVR_BENCH_SAME_DEVICE=1; the line says `shared_device`): the control flow, not a scaling measurement."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    if n_devices < n:
        env["REEVR_BENCH_SAME_DEVICE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def write_full(full: dict, where: str = ""):
    """The full record (tens of KB) goes to a file, never to the line the driver parses."""
    outs = [where] if where else [os.path.join(ROOT, "bench_full.json")]
    if not where and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        outs.append(os.path.join(ROOT, "gpurun_out", "bench_full.json"))
    done = []
    for o in outs:
        try:
            with open(o, "w") as f:
                json.dump(full, f, indent=1)
            done.append(os.path.relpath(o, ROOT) if o.startswith(ROOT) else o)
        except OSError:
            pass
    return done


# kernel family -> a substring of the rocprofv3 kernel name (the committed kernel-trace summaries list C++ names)
FAMILY_KERNEL = {"fused_block": "k_fused_block", "fir_tail": "k_fdl_patch", "fir_head": "k_fdl_patch<0", "fft_fwd_tail": "k_fft8_fwd",
                 "fft_inv_tail": "k_fft8_inv"}


def rocprof_cross_check(roof: dict, cfg: int, one_queue: bool) -> dict:
    """HISTORICAL annotation, full record only (never the parsed line): the dominant kernel's average duration in the COMMITTED
    `rocprofv3 --kernel-trace --stats` summary of this command (profiles/<PROFILE_ROUND>_config<C>/kernel_stats*.csv -- this
    round's only, no fallback to an older build's profile) and the fraction of the HBM peak it gives with this run's bytes per
    launch. `frac` beside it is this run's own HIP-event measurement; after a kernel change the two differ until the profile
    is taken again (tools/profile_configs.sh)."""
    import csv
    pat = FAMILY_KERNEL.get(roof.get("kernel", ""))
    if not pat or not roof.get("bytes_per_launch"):
        return {}
    f = os.path.join(ROOT, "profiles", "%s_config%d" % (PROFILE_ROUND, cfg), "kernel_stats_one_queue.csv" if one_queue else "kernel_stats.csv")
    if not os.path.exists(f):
        return {}
    rows = [r for r in csv.DictReader(open(f)) if pat in r["Name"]]
    if not rows:
        return {}
    r = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    avg_ms = float(r["AverageNs"]) * 1e-6
    return {"rocprof_committed": {"avg_launch_ms": round(avg_ms, 5),
                                  "frac": round(roof["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "summary": os.path.relpath(f, ROOT),
                                  "note": "committed profile of an earlier run of this command, not this run's measurement"}}


COMPACT_LIMIT = 4096


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def compact_line(full: dict, full_path=None) -> dict:
    """The ONE line the driver parses, from the full record: the contract's keys, `roofline` of the dominant kernel,
    `cpu_baseline`, the probe, and a handful of scalars per other configuration / regime. Always < COMPACT_LIMIT bytes:
    optional groups are dropped (never the contract's keys) should a run produce more."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data"))
    cfg = full.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "baseline_config", "channels_per_gpu", "instances_total", "frames_per_channel_per_step",
                                 "host_block", "calls_per_step", "partitions", "tail_stage", "transforms", "tail_block_run", "tile_blocks", "subsets",
                                 "resident_GB", "schedule", "gather", "gathered_channels_per_gpu", "gather_matches_output", "devices",
                                 "shared_device", "tune", "plan_as_tested"))
    if len(str(line["config"].get("workload", ""))) > 200:
        line["config"]["workload"] = line["config"]["workload"][:197] + "..."
    roof = full.get("roofline")
    if roof:
        r = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_launch_ms",
                         "traffic_over_model", "alg_frac_reference_schedule", "alg_equiv", "frac_whole_step_executed_bytes",
                         "default_run_frac"))
        for k in ("traffic_source", "measured_in"):
            if roof.get(k):
                r[k] = str(roof[k])[:120]
        line["roofline"] = r
    else:
        line["roofline"] = None
    pr = full.get("probe")
    line["probe"] = {"ok": pr["ok"], "rms_error": float("%.3g" % pr["rms_error"])} if pr else None
    cb = full.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "cpu_model"))
        c["sample"] = str(cb.get("sample", ""))[:160]
        # (the all-cores figure -- a fixed wall budget on a shared host, 17-40x spread between threads -- stays in the full record only)
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    optional = []
    others = {}
    for key in ("config1", "config3", "config5"):
        ent = full.get(key)
        if ent:
            others[key[-1]] = {"Msamples_s": ent["value"], "exec_frac": ent.get("frac_of_hbm_peak_executed_bytes"),
                               "ref_schedule_alg_frac": ent.get("alg_frac_reference_schedule"),
                               "one_queue_Msamples_s": (ent.get("one_queue") or {}).get("value"),
                               "cpu_1thread_Msamples_s": (ent.get("cpu_baseline") or {}).get("value"),
                               "probe_ok": bool(ent.get("probe") and ent["probe"]["ok"])}
    if others:
        line["other_configs"] = others
        optional.append("other_configs")
    side = {}
    cu = full.get("call_us")
    if cu:                                  # what each per-block call of the timed set costs the device (stamped loop)
        line["call_us"] = _pick(cu, ("p50", "p99", "max", "block_period_us", "max_over_block_period", "tail_phase_groups"))
    if (full.get("bg_stream") or {}).get("call_us"):
        side["bg_stream_Msamples_s"] = full["bg_stream"]["value"]
        side["bg_stream_call_us_p99_max"] = [full["bg_stream"]["call_us"]["p99"], full["bg_stream"]["call_us"]["max"]]
    if full.get("reference_schedule"):
        side["reference_schedule_Msamples_s"] = full["reference_schedule"]["value"]
    if full.get("one_queue"):
        side["one_queue_Msamples_s"] = full["one_queue"]["value"]
    sb = full.get("stereo_block_sync")
    if sb:
        side["stereo_pair_us_per_block"] = sb["us_per_block"]
        side["stereo_pair_host_call_us_median"] = sb["host_call_us_median"]
    hb = full.get("host_boundary")
    if hb:                                  # PCIe-inclusive rate of the host-pointer boundary at 1024 / 4096 channels: [own buffers, in place]
        side["host_pcie_inclusive_Msamples_s"] = {k: [v["own_buffers_Msamples_s"], v["in_place_Msamples_s"]] for k, v in hb.items() if isinstance(v, dict)}
    reg = full.get("regimes")
    if reg:
        sw = reg.get("channel_sweep", {})
        for ch in ("2", "16", "64", "256", "1024"):
            if isinstance(sw.get(ch), dict):
                side["ch%s_Msamples_s" % ch] = sw[ch]["value"]
                side["ch%s_us_per_block" % ch] = sw[ch]["us_per_block"]
        for k in ("config5_literal", "fft_f32", "fft_f64_long", "fft_f64"):
            if k in reg:
                side[k + "_Msamples_s"] = reg[k]["value"]
        r5 = (reg.get("config5_literal") or {}).get("roofline")
        if r5:
            side["config5_literal_roofline"] = dict(_pick(r5, ("bound", "achieved", "frac", "unit")),
                                                    mac_frac_fp32=(r5["families"].get("fir_tail") or {}).get("frac_fp32"))
    if side:
        line["side"] = side
        optional.insert(0, "side")
    if full_path:
        line["full_record"] = full_path
    for k in optional + ["full_record"]:                 # never reached by a normal run: shed optional groups, largest last
        if len(json.dumps(line)) < COMPACT_LIMIT:
            break
        line.pop(k, None)
    if len(json.dumps(line)) >= COMPACT_LIMIT:           # a pathological workload string or partitions table
        line["config"] = _pick(line["config"], ("workload", "baseline_config", "channels_per_gpu", "host_block"))
        line["config"]["workload"] = str(line["config"].get("workload", ""))[:80]
    assert len(json.dumps(line)) < COMPACT_LIMIT
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", type=int, default=2, choices=(1, 2, 3, 4, 5), help="BASELINE.json configuration (1-based index)")
    ap.add_argument("--lockstep", type=int, default=0, help="with --config 5: 1 = config 5's GEOMETRY (5 s IR, block 4096 -> head 4096 / tail 8192) in the "
                    "lock-step regime as the headline (the entry `config5` of the default line) instead of the literal 64-channel offline render")
    ap.add_argument("--channels", type=int, default=0, help="lock-step channels per GPU (2 per stereo instance; 0: the config's default)")
    ap.add_argument("--time-tiling", type=int, default=1, help="0: RVC_FLAG_NO_TIME_TILING (the reference's per-block sweep order)")
    ap.add_argument("--blocks-per-step", type=int, default=0, help="block-synchronous configs: host blocks per step (0: the config's default)")
    ap.add_argument("--bg-stream", type=int, default=0, help="1: tail stage on the second HIP stream")
    ap.add_argument("--gather", type=int, default=1,
                    help="under torch.distributed.run: RCCL all_gather per step; 1: configs 4/5 every channel, others the "
                         "outputs of 8 stereo instances per GPU; 2: every channel; 0: off")
    ap.add_argument("--configs", type=str, default="1,3,5", help="other BASELINE configurations measured in the same run (N = 1, config 2)")
    ap.add_argument("--config-steps", type=int, default=8, help="timed steps of each of those")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the headline's CPU baseline leg (0 = skip)")
    ap.add_argument("--config-cpu-seconds", type=float, default=4.0, help="budget of each other configuration's CPU leg")
    ap.add_argument("--side", type=int, default=1, help="0: skip the side measurements")
    ap.add_argument("--regimes", type=int, default=1, help="0: skip the small-regime entries (channel sweep, literal config 5, double transforms)")
    ap.add_argument("--distinct", type=int, default=0, help="synthesise only this many different stereo IRs and cycle them (0: all different; config 3: 128)")
    ap.add_argument("--child-sets", type=int, default=1, help="0: RVC_FLAG_NO_SUBSETS for the measured set (one set on one queue; the "
                    "default serves thousands of block-synchronous channels by child sets on their own streams, fenced internally)")
    ap.add_argument("--ir-len", type=int, default=0, help="measurement hook: impulse length in samples instead of the configuration's (the "
                    "line then is NOT the BASELINE configuration: config.workload says so)")
    ap.add_argument("--tune", type=str, default="", help="rvc_debug_set_tuning knobs, e.g. k1=32,subsets=2 (measurement hook)")
    ap.add_argument("--full-out", type=str, default="", help="where the full record goes (default: bench_full.json beside bench.py, "
                    "and gpurun_out/bench_full.json where that directory exists)")
    ap.add_argument("--watchdog", type=float, default=1500.0,
                    help="seconds after which a stuck run dumps every thread's stack and exits (0: off)")
    args = ap.parse_args()
    wall0 = time.perf_counter()
    wall = {}

    def lap(name):                 # wall-clock seconds since the start of main() at the end of each phase (full record: `wall_s`)
        wall[name] = round(time.perf_counter() - wall0, 1)
    if args.watchdog > 0:          # a hung collective or kernel must not hold the GPU box until the driver's limit
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

    import torch
    import reevr_amd
    from reevr_amd import KERNEL_NAMES, shard, synth
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        assert reevr_amd.set_tuning(k, int(v)), k
    if args.ir_len > 0:                      # (measurement hook: another impulse length at the configuration's geometry)
        WORKLOADS[args.config] = dict(WORKLOADS[args.config], ir_len=args.ir_len,
                                      text=WORKLOADS[args.config]["text"] + " -- NOT the BASELINE configuration: --ir-len %d" % args.ir_len)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus, torch.cuda.device_count())        # does not return: the ranks print the line
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: a launcher started {world} ranks for a {args.gpus}-GPU run "
                         f"(bare `python bench.py --gpus {args.gpus} ...` starts its own ranks)")
    # all ranks on GPU 0 over gloo: the N > 1 control flow on a box with fewer devices than ranks (self_launch sets it there)
    same_device = os.environ.get("REEVR_BENCH_SAME_DEVICE") == "1"
    devices = [0] * world if same_device else list(range(world))
    if same_device:
        local_rank = 0
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # under torch.distributed.run (also with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if same_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- workload ----------------------------------------------------------------------------
    long_call = False
    wcfg = args.config
    if args.config in (1, 2, 3) or (args.config == 5 and args.lockstep):
        w = WORKLOADS[args.config]
        channels = args.channels or w["channels"]
        if channels < 2 or channels % 2:
            raise SystemExit("--channels must be a positive even number (stereo instances)")
        n_inst = channels // 2
        instances = shard.units_for_rank(n_inst * world, world, rank)  # rank = unit mod world, equal shards
        scaling = "weak"
        workload = f"{w['text']}: {n_inst} stereo instances per GPU in lock-step, one process() per {w['host_block']}-frame block"
    elif args.config == 4:
        wcfg = 2
        if 8 % world:
            raise SystemExit("--config 4 shards 8 stereo instances: --gpus must divide 8")
        instances = shard.units_for_rank(8, world, rank)
        scaling = "strong"
        workload = "8 independent stereo instances, 10 s IR @ 48 kHz, block=512, sharded over the GPUs (unit mod world)"
    else:
        long_call = True
        if 32 % world:
            raise SystemExit("--config 5 shards 64 mono channels (32 pairs): --gpus must divide 32")
        instances = shard.units_for_rank(32, world, rank)
        scaling = "strong"
        workload = "batched offline render: 64 mono channels, 5 s IR @ 48 kHz, block=4096, 64/N channels per GPU, one long call per step"
    blocks = args.blocks_per_step or WORKLOADS[wcfg]["blocks"]
    head, tail = geometry(WORKLOADS[wcfg]["host_block"], WORKLOADS[wcfg]["single"])
    host_block = WORKLOADS[wcfg]["host_block"]
    if not long_call and (blocks < 1 or (blocks * host_block) % (tail or head)):
        raise SystemExit("--blocks-per-step must cover whole tail periods (a multiple of %d)" % ((tail or head) // host_block))

    ls = Lockstep(torch, reevr_amd, synth, wcfg, instances, local_rank, bool(args.time_tiling), bool(args.bg_stream), blocks,
                  long_call=long_call, distinct=args.distinct or (128 if wcfg == 3 else 0), child_sets=None if args.child_sets else False)
    conv, nch, frames_step, nbuf, ir_len = ls.conv, ls.nch, ls.frames_step, ls.nbuf, ls.ir_len
    do_gather = bool(args.gather and dist is not None)
    # The output batch of every step is gathered with ONE all_gather, overlapped with the next step's compute:
    # the collective is ordered behind this step's kernels (torch's current stream waits for the set's streams) and runs
    # on the communicator's stream; a batch buffer is reused only after its own gather has completed.
    gch = nch if (args.gather >= 2 or args.config == 4 or long_call) else min(nch, 16)      # channels per rank that are gathered
    g_out = [torch.empty((world,) + (gch, frames_step), dtype=torch.float32, device=dev) for _ in range(nbuf)] if do_gather else None
    g_stage = [torch.empty((gch, frames_step), dtype=torch.float32, device=dev) for _ in range(nbuf)] if do_gather else None
    pending = [None] * nbuf

    def step(gather=True):
        if do_gather and gather:
            b = ls.i % nbuf
            if pending[b] is not None:         # the gather that read this batch's buffers two steps ago
                pending[b].wait()
                pending[b] = None
            # (every channel gathered: the set writes the contiguous staging batch the collective reads in place)
            _, yo = ls.step(out=g_stage[b] if gch == nch else None, order=True)
            if gch != nch:
                g_stage[b].copy_(yo[:gch])     # (torch's current stream, ordered behind the set's streams by order=True)
            pending[b] = shard.gather_batches_async(g_stage[b], g_out[b], dist)
            return
        ls.step()

    def drain():
        for b in range(nbuf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def fence():
        drain()
        conv.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    lap("import_synth_init")
    # the plan this set runs against the one the steady-state parity tests cover (None = the same; literal config 5: no table entry)
    plan_diff = None if long_call else plan_as_tested(conv, wcfg)
    pre = ls.preroll()
    ls_period = ls.tile_period_steps()     # (a fair average needs --steps to be a multiple of this: 1 except for --config 3, where it is 4)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    # closing bracket: synchronise this rank, stamp, barrier; the reported time is the MAX over ranks of
    # the stamped spans (all ranks left the opening barrier together)
    drain()
    conv.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    conv.check()
    probe = ls.check_probe()                  # (channel 0's last timed step against its impulse response)
    gather_ok = None
    if do_gather:                              # the gathered batch of the last step against the set's own output
        lb = (ls.i - 1) % nbuf
        mine = g_out[lb][rank if not same_device else dist.get_rank()]
        _, yo = ls.batch(lb)
        src = g_stage[lb] if gch == nch else yo[:gch]
        gather_ok = bool(torch.equal(mine, src))
    elapsed = shard.max_over_ranks(elapsed, dist, dev)    # slowest rank
    total_ch = nch * world if (args.config in (1, 2, 3) or not long_call and args.config == 5) else (16 if args.config == 4 else 64)
    total_samples = total_ch * frames_step * args.steps
    value = total_samples / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    lap("timed_run")
    # ---- per-kernel durations, live, with HIP events on the streams the kernels run on ----
    kern = ls.kernel_times(KERNEL_NAMES)
    call_us = ls.call_latency() if (not long_call and world == 1) else None     # cost of each per-block call, stamped on the device
    PA, PT = conv.partitions(0), conv.partitions(1)
    transforms = transforms_form(conv)
    tail_x = ls.tail_used                 # the tail block the set runs (twice the requested one where the engine widens long tails)
    tail_form = tail_stage_form(conv, head, tail)
    tiled = ls.tiled
    exe = {} if long_call else executed_bytes(conv, nch, head, tail, ir_len, host_block, tiled)
    traffic_all, tsrc = ({}, None) if long_call else load_traffic(nch // max(1, conv.subsets), args.config, tiled)
    roof_all, exe_bytes_step = roofline_tables(kern, exe, traffic_all)
    bps = alg_bytes_per_sample(head, tail, ir_len)
    fps = flops_per_sample(head, tail, ir_len)
    rate_gpu = value / world * 1e6
    roof = None
    if roof_all:
        dominant = max(roof_all, key=lambda k: roof_all[k]["ms_per_step"])      # largest share of a step
        r = roof_all[dominant]
        roof = {"bound": "hbm", "kernel": dominant, "achieved": r["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": r["frac"], "traffic": r["traffic"], "bytes_per_launch": r["bytes_per_launch"],
                "avg_launch_ms": r["avg_launch_ms"], "traffic_source": tsrc,
                # child sets (the default for thousands of channels): launches of the family run side by side, `achieved` = bytes
                # over the UNION of the launch intervals; frac_per_launch / avg_launch_ms are the literal per-launch figures. On one
                # queue (--child-sets 0, side entry one_queue) every launch has the device to itself and the two coincide
                "concurrency": r["concurrency"], "frac_per_launch": r["frac_per_launch"], "busy_ms_per_step": r["ms_per_step"],
                # SURVEY.md 8d's numerator (1497 B per channel-sample of the REFERENCE's loop nest): as a fraction of the
                # peak only where the executed schedule moves those bytes (reference_schedule, filled in below); with
                # time tiling the same figure x this run's rate is a throughput EQUIVALENT (> 1 is not an efficiency)
                "alg_frac_reference_schedule": None,
                "alg_equiv": round(rate_gpu * bps / 1e9 / HBM_PEAK_GBS, 4),
                "alg_bytes_per_sample": round(bps, 1),
                "note": ("time-tiled block-synchronous schedule: bytes_per_launch = the bytes of the structure this launch executes "
                         "(a sweep reads the stage's IR spectra and arrived delay-line rows ONCE per tile; a patch / the per-block "
                         "launch only the partitions that arrived since), averaged over the launches of its family; frac <= 1. "
                         "alg_frac_reference_schedule = SURVEY 8d bytes x the rate of the reference-order run / 8 TB/s; alg_equiv = "
                         "the same bytes x THIS run's rate / 8 TB/s (a throughput equivalent, not an efficiency)"
                         if tiled else
                         "reference schedule: every launch re-reads the IR spectra and the delay line of its stage once (16 B per "
                         "partition x bin), physical HBM bytes = SURVEY.md 8d algorithmic bytes, frac <= 1") +
                        "; 0.5-4 flop/B, far below the fp32 ridge"}
        if not tiled:
            roof["alg_frac_reference_schedule"] = roof["alg_equiv"]
    exe_bps = exe_bytes_step / (nch * frames_step) if exe_bytes_step else None
    path = {"executed_bytes_per_sample": round(exe_bps, 1) if exe_bps else None,
            "executed_GBs_per_gpu": round(rate_gpu * exe_bps / 1e9, 1) if exe_bps else None,
            "frac_of_hbm_peak": round(rate_gpu * exe_bps / 1e9 / HBM_PEAK_GBS, 4) if exe_bps else None,
            "reference_alg_bytes_per_sample": round(bps, 1),
            "reference_alg_GBs_equivalent": round(rate_gpu * bps / 1e9, 1),
            "reference_alg_frac_equivalent": round(rate_gpu * bps / 1e9 / HBM_PEAK_GBS, 4),
            "flops_per_sample": round(fps, 1), "frac_of_fp32_peak": round(rate_gpu * fps / 1e12 / FP32_PEAK_TFLOPS, 4),
            "x_realtime_per_gpu": round(rate_gpu / SR, 1),
            "note": "frac_of_hbm_peak = bytes the executed schedule moves per channel-sample x measured rate / 8 TB/s (<= 1). "
                    "reference_alg_*: SURVEY.md 8d's bytes of the REFERENCE's loop nest x the same rate -- with time tiling "
                    "this exceeds the physical traffic by the tiling's byte saving and is a throughput equivalent, not an "
                    "efficiency; the `reference_schedule` entry is the run where the two coincide."}

    side, cpu, others, otsrc = {}, None, {}, None
    lockstep_cfg = args.config in (1, 2, 3) or (args.config == 5 and not long_call)
    irs, x = ls.irs, ls.x
    init_ms, synth_s, subsets = ls.init_ms, ls.synth_s, conv.subsets
    tiles = {"zero-latency stage": conv.tile_rows(0), "tail stage": conv.tile_rows(1)}
    if args.side and lockstep_cfg and world == 1:
        ls.close()
        if tiled:       # the same loop in the reference's sweep order (same channels, same inputs)
            rs = Lockstep(torch, reevr_amd, synth, wcfg, instances, local_rank, False, bool(args.bg_stream), blocks, irs=irs, x=x)
            rs.preroll()
            rsteps = max(2, args.steps // 4)
            rrate, rms = rs.timed(rsteps, 1)
            rs.conv.check()
            rs.close()
            side["reference_schedule"] = {
                "value": round(rrate / 1e6, 3), "unit": "Msamples/s", "steps": rsteps, "ms_per_step": round(rms, 4),
                "alg_bytes_per_sample": round(bps, 1), "achieved_GBs": round(rrate * bps / 1e9, 1),
                "frac_of_hbm_peak": round(rrate * bps / 1e9 / HBM_PEAK_GBS, 4),
                "note": "RVC_FLAG_NO_TIME_TILING: same channels / inputs / call pattern, every host block sweeps every "
                        "partition (FFTConvolver.cpp:176-187): physical bytes = SURVEY.md 8d algorithmic bytes"}
            if roof is not None:
                roof["alg_frac_reference_schedule"] = side["reference_schedule"]["frac_of_hbm_peak"]
        if subsets > 1:
            # the same loop on ONE queue (RVC_FLAG_NO_SUBSETS): every launch has the device to itself, so per family bytes per
            # launch / mean launch duration is an efficiency -- the per-kernel figure a rocprofv3 kernel-stats average corresponds to
            oq = Lockstep(torch, reevr_amd, synth, wcfg, instances, local_rank, True, bool(args.bg_stream), blocks, irs=irs, x=x,
                          child_sets=False)
            oq.preroll()
            osteps = max(2, args.steps // 2)
            orate, oms = oq.timed(osteps, 1)
            oprobe = oq.check_probe()
            oq.conv.check()
            okern = oq.kernel_times(KERNEL_NAMES)
            oexe = executed_bytes(oq.conv, nch, head, tail, ir_len, host_block, oq.tiled)
            otraffic, otsrc = load_traffic(nch, args.config, oq.tiled)
            oroof, _ = roofline_tables(okern, oexe, otraffic)
            oq.close()
            side["one_queue"] = {
                "value": round(orate / 1e6, 3), "unit": "Msamples/s", "steps": osteps, "ms_per_step": round(oms, 4),
                "probe_ok": bool(oprobe and oprobe["ok"]),
                "roofline_all": {k: {"launches_per_step": v["launches_per_step"], "avg_launch_ms": v["avg_launch_ms"],
                                     "bytes_per_launch": v["bytes_per_launch"], "frac": v["frac"], "traffic": v["traffic"]}
                                 for k, v in oroof.items()},
                "note": "RVC_FLAG_NO_SUBSETS run of the headline loop (same channels, inputs, call pattern): one set on one queue; "
                        "frac = bytes per launch / mean launch duration / 8 TB/s"}
        if not args.bg_stream and tail:
            # the reference's own structure: the tail job on a second stream (RVC_FLAG_BG_STREAM, delay 2, Convolver.cpp:84-95) -- rate
            # and what each call costs the FOREGROUND stream
            bgl = Lockstep(torch, reevr_amd, synth, wcfg, instances, local_rank, True, True, blocks, irs=irs, x=x)
            bgl.preroll()
            brate, bms = bgl.timed(max(2, args.steps // 4), 1)
            bprobe = bgl.check_probe()
            bgl.conv.check()
            blat = bgl.call_latency()
            side["bg_stream"] = {"value": round(brate / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(bms, 4),
                                 "probe_ok": bool(bprobe and bprobe["ok"]), "call_us": blat, "subsets": bgl.conv.subsets,
                                 "tail_stage": tail_stage_form(bgl.conv, head, tail),
                                 "note": "RVC_FLAG_BG_STREAM: the reference's stage split, tail job on a second HIP stream; call_us = "
                                         "completion stamps on the foreground streams"}
            bgl.close()
        if args.config == 2:
            side.update(side_measurements(torch, reevr_amd, synth, irs[:2], local_rank, dev, host_block, tail))
            side["host_boundary"] = host_boundary(torch, reevr_amd, irs, x, local_rank, host_block, tail)
    lap("side_legs")
    if world == 1 and args.cpu_seconds > 0 and lockstep_cfg:
        cores = os.cpu_count() or 1
        n_cpu_irs = min(len(irs), max(2, cores))
        xin = [np.ascontiguousarray(x[1 + c % (nch - 1)]) for c in range(n_cpu_irs)]          # (channel 0 carries the probe)
        cpu = cpu_baseline(irs[:n_cpu_irs], xin, host_block, tail, args.cpu_seconds, WORKLOADS[wcfg]["text"])
    lap("cpu_baseline")
    regimes = None
    if args.config == 2 and world == 1 and args.side and args.regimes:
        regimes = small_regimes(torch, reevr_amd, synth, KERNEL_NAMES, local_rank, 4, irs4096=irs, x4096=x, instances=instances)
    lap("regimes")
    if args.config == 2 and world == 1 and args.side:
        del irs, x
        for c in [int(v) for v in args.configs.split(",") if v.strip()]:
            if c in WORKLOADS and c != 2:
                others["config%d" % c] = side_config(torch, reevr_amd, synth, KERNEL_NAMES, c, WORKLOADS[c]["channels"],
                                                     local_rank, args.config_steps, args.config_cpu_seconds)

    lap("other_configs")
    # Compact scalar summary of everything the line holds elsewhere in nested form (the driver keeps the scalar entries of
    # `config` / `roofline`): per side configuration its rate, executed-bytes fraction of the HBM peak, SURVEY 8d fraction of
    # the reference-order run and 1-thread CPU rate; the small regimes; the one-queue run.
    summary = {}
    for key, ent in others.items():
        c = key.replace("config", "c")
        summary[c + "_Msamples_s"] = ent["value"]
        summary[c + "_exec_frac"] = ent["frac_of_hbm_peak_executed_bytes"]
        summary[c + "_alg_frac_ref_schedule"] = ent["alg_frac_reference_schedule"]
        summary[c + "_one_queue_Msamples_s"] = ent.get("one_queue", {}).get("value")
        summary[c + "_cpu_1thread_Msamples_s"] = (ent.get("cpu_baseline") or {}).get("value")
        summary[c + "_probe_ok"] = bool(ent["probe"] and ent["probe"]["ok"])
    if regimes:
        for ch, ent in regimes["channel_sweep"].items():
            if isinstance(ent, dict):
                summary["ch%s_Msamples_s" % ch] = ent["value"]
                summary["ch%s_us_per_block" % ch] = ent["us_per_block"]
                summary["ch%s_exec_frac" % ch] = ent["frac_of_hbm_peak_executed_bytes"]
        summary["config4_literal_1gpu_Msamples_s"] = regimes["channel_sweep"]["16"]["value"]
        summary["config5_literal_Msamples_s"] = regimes["config5_literal"]["value"]
        for k in ("fft_f32", "fft_f64_long", "fft_f64"):
            if k in regimes:
                summary[k + "_Msamples_s"] = regimes[k]["value"]
    if "one_queue" in side:
        summary["one_queue_Msamples_s"] = side["one_queue"]["value"]
        oq = side["one_queue"]["roofline_all"].get(roof["kernel"]) if roof else None
        if oq:
            # The headline `value` is the DEFAULT run, whose child sets run two launches of a family (and of other families) side
            # by side: there a launch's duration depends on what else is running, and bytes per launch / mean launch duration is not an
            # efficiency. The contract's per-launch roofline of the dominant kernel therefore comes from the `one_queue` leg of
            # this same run (RVC_FLAG_NO_SUBSETS: every launch has the device to itself; its rocprofv3 average:
            # profiles/r4_config2/kernel_stats_one_queue.csv); the default run's own figures stay beside it as default_run_*
            # (bytes over the UNION of the family's launch intervals; rocprofv3 side: profiles/r4_config2/kernel_union.txt).
            for k in ("achieved", "frac", "traffic", "bytes_per_launch", "avg_launch_ms", "concurrency", "frac_per_launch",
                      "busy_ms_per_step", "traffic_source"):
                roof["default_run_" + k] = roof[k]
            roof.update({"achieved": round(oq["bytes_per_launch"] / (oq["avg_launch_ms"] * 1e-3) / 1e9, 1), "frac": oq["frac"],
                         "traffic": oq["traffic"], "bytes_per_launch": oq["bytes_per_launch"], "avg_launch_ms": oq["avg_launch_ms"],
                         "concurrency": 1.0, "frac_per_launch": oq["frac"],
                         "busy_ms_per_step": round(oq["avg_launch_ms"] * oq["launches_per_step"], 5),
                         "traffic_source": otsrc, "measured_in": "one_queue leg of this run (RVC_FLAG_NO_SUBSETS, same channels / inputs / "
                                                                 "call pattern); value / ms_per_step: the default run (child sets)"})
    if roof is not None:
        roof.update(rocprof_cross_check(roof, args.config, one_queue="default_run_frac" in roof))
        roof["probe_ok"] = bool(probe and probe["ok"])
        roof["frac_whole_step_executed_bytes"] = path["frac_of_hbm_peak"]
        if roof.get("traffic") and roof.get("bytes_per_launch"):
            roof["traffic_over_model"] = round(roof["traffic"] / roof["bytes_per_launch"], 4)
        if roof.get("default_run_traffic") and roof.get("default_run_bytes_per_launch"):
            roof["default_run_traffic_over_model"] = round(roof["default_run_traffic"] / roof["default_run_bytes_per_launch"], 4)

    full = {
        "metric": "Msamples/s convolved (stereo, 10s IR, block=512); % HBM roofline",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "baseline_config": args.config,
                   "channels_per_gpu": nch, "stereo_instances_per_gpu": nch // 2, "instances_total": total_ch // 2,
                   "frames_per_channel_per_step": frames_step, "host_block": host_block,
                   "calls_per_step": 1 if long_call else frames_step // host_block,
                   "partitions": {"zero-latency stage (block %d)" % head: PA, "tail stage (block %d)" % tail_x: PT},
                   "tail_stage": tail_form, "transforms": transforms,
                   "tail_block_requested": tail, "tail_block_run": tail_x,
                   "tile_blocks": tiles, "subsets": subsets,
                   "resident_GB": round(nch * 8.0 * (PA * head + (PT + 2) * tail_x) * 2 / 1e9, 2),
                   "schedule": "long call" if long_call else ("causal time tiling (sweeps + patches; two levels for long delay lines)"
                                                                if tiled else "reference order (RVC_FLAG_NO_TIME_TILING)"),
                   "call": ("one process() per step" if long_call else
                            "one process_device() per %d-frame host block for all channels (rvc_set_process_device_blocks), "
                            "device-resident I/O, %d input/output batches rotated" % (host_block, nbuf)),
                   "tile_period_steps": ls_period, "pre_roll_steps": pre, "gather": do_gather, "gathered_channels_per_gpu": gch if do_gather else 0,
                   "gather_matches_output": gather_ok, "tune": args.tune,
                   "plan_as_tested": (plan_diff is None) if not long_call else None, "plan_diff": plan_diff,
                   "sharding": "instances dealt to ranks, equal shards, no data-path collective"
                               + (f"; one RCCL all_gather of the output blocks of {gch} channels per GPU per step, overlapped with the next step" if do_gather else "")},
        "roofline": roof,
        "probe": probe,
        "call_us": call_us,
        "roofline_all": roof_all,
        "path_roofline": path,
        "kernels_ms": {k: round(v["avg_ms"], 5) for k, v in kern.items()},
        **side,
        "regimes": regimes,
        "summary": summary,
        "cpu_baseline": cpu,
        **others,
        "init_ms": round(init_ms, 2), "synth_s": round(synth_s, 2), "wall_s": wall,
    }
    full["config"]["devices"] = devices
    full["config"]["shared_device"] = same_device
    # The whole record goes to a side file (its path on an EARLIER stdout line); the LAST stdout line is the compact
    # (< 4 KB) line the driver parses: contract keys, roofline, cpu_baseline, probe and a dozen scalars of the other configs.
    paths = write_full(full, args.full_out)
    print("bench.py: full record (per-family rooflines, regimes, the other configurations) -> " + ", ".join(paths), flush=True)
    print(json.dumps(compact_line(full, paths[0] if paths else None)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if probe is not None and not probe["ok"]:
        raise SystemExit("correctness probe failed: %r" % (probe,))


def host_boundary(torch, reevr_amd, irs, x, local_rank, host_block, tail, channels=(1024, 4096), blocks=192):
    """The host-pointer boundary with MANY channels, PCIe-inclusive (never `value`), in steady state (the delay lines are filled
    through the device entry first): rvc_set_process per block on the caller's own buffers (staging copy by the copy crew + DMA
    both ways) and `in_place` -- the caller's audio lives in the set's pinned staging rows (rvc_set_host_buffers), both through
    the same C loop with a stopwatch around every call (whole tail periods: every 16th call carries the tail job)."""
    out = {}
    for nch in channels:
        if nch > len(irs):
            continue
        s = reevr_amd.ConvolverSet(nch, device=local_rank)
        assert s.init(host_block, tail, irs[:nch], max_len=host_block), s.last_error_string
        pre = -(-(s.partitions(1) + 4) * int(s.tail_block) // x.shape[1])
        dx = torch.from_numpy(np.ascontiguousarray(x[:nch])).to(torch.device("cuda", local_rank))
        dy = torch.empty_like(dx)
        for _ in range(pre):
            s.process_device_blocks(dx, host_block, dy, sync=True)
        del dx, dy
        torch.cuda.empty_cache()
        xs = np.ascontiguousarray(x[:nch, :host_block * blocks])
        _, us = s.process_host_blocks_timed(xs, host_block)
        s.check()
        ins, outs = s.host_buffers()
        for c in range(nch):
            ins[c][:host_block] = xs[c, :host_block]
        us2 = s.process_in_place_timed(host_block, blocks)
        s.check()
        rate = lambda u: round(nch * host_block * len(u) / (float(np.sum(u)) * 1e-6) / 1e6, 1)
        q = lambda u: [round(float(np.sort(u)[len(u) // 2]), 1), round(float(np.sort(u)[int(len(u) * 0.99)]), 1)]
        out[str(nch)] = {"own_buffers_Msamples_s": rate(us), "in_place_Msamples_s": rate(us2),
                         "own_buffers_call_us_p50_p99": q(us), "in_place_call_us_p50_p99": q(us2),
                         "MB_per_call_each_way": round(nch * host_block * 4 / 1e6, 2), "subsets": s.subsets, "pre_roll_blocks": pre * (x.shape[1] // host_block)}
        s.close()
    out["note"] = ("rvc_set_process on HOST buffers, one call per block, PCIe-inclusive, delay lines full: own_buffers = the caller's "
                   "per-channel buffers (staged into pinned rows by a few host threads, DMA both ways), in_place = the caller writes / "
                   "reads the set's pinned staging rows (rvc_set_host_buffers); every 16th call carries the tail job")
    return out


def side_measurements(torch, reevr_amd, synth, irs2, local_rank, dev, host_block, tail):
    """ONE stereo pair (the plug-in's own case), three ways. Not the headline."""
    out = {}
    # (a) block-synchronous: per 512-frame block, device-resident loop and host-pointer calls (the audio thread's view),
    #     with one launch per block (tail job inline / on the second stream)
    nblk = 3000
    xs_h = np.stack([synth.synth_input(host_block * nblk, c) for c in range(2)])
    xs = torch.from_numpy(xs_h).to(dev)
    ys = torch.empty_like(xs)
    torch.cuda.synchronize()
    res = {}
    for mode, kw in (("tail_on_second_stream", dict(bg_stream=True)), ("tail_inline", dict(bg_stream=False)),
                     ("tail_inline_f32", dict(bg_stream=False, fft_f32=True))):
        s = reevr_amd.ConvolverSet(2, device=local_rank, **kw)
        assert s.init(host_block, tail, irs2, max_len=host_block)
        s.process_device_blocks(xs[:, :host_block * 200].contiguous(), host_block)
        ts = time.perf_counter()
        s.process_device_blocks(xs, host_block, ys)
        te = time.perf_counter() - ts
        _, us = s.process_host_blocks_timed(xs_h[:, :host_block * 1500], host_block)
        us = np.sort(us[300:])
        res[mode] = {"Msamples_s": round(2 * host_block * nblk / te / 1e6, 3), "us_per_block": round(te / nblk * 1e6, 2),
                     "host_call_us_median": round(float(us[len(us) // 2]), 2), "host_call_us_p99": round(float(us[int(len(us) * 0.99)]), 2)}
        s.close()
    best = max(res.values(), key=lambda r: r["Msamples_s"])
    out["stereo_block_sync"] = {"value": best["Msamples_s"], "unit": "Msamples/s", "us_per_block": best["us_per_block"],
                                "host_call_us_median": min(r["host_call_us_median"] for r in res.values()),
                                "modes": res,
                                "note": "ONE stereo pair. us_per_block: one process_device() call per 512-frame block back to back "
                                        "(host loop in C); host_call_us: rvc_set_process() on host buffers per block, back to back "
                                        "(pinned staging + hand-off + kernel + copy back; stopwatch in C). Default precision of a "
                                        "set this small: tail transforms (8192) in double; tail_inline_f32 = RVC_FLAG_FFT_F32."}
    # (b) offline: one 40 s call per step (adaptive partitioning) and (c) the same through the fixed head/tail sizes
    frames = 40 * SR
    xl = torch.from_numpy(np.stack([synth.synth_input(frames, c) for c in range(2)])).to(dev)
    yl = torch.empty_like(xl)
    for key, fixed in (("stereo_offline_long_call", False), ("stereo_offline_fixed_partitions", True)):
        s = reevr_amd.ConvolverSet(2, device=local_rank, fixed_partitions=fixed)
        assert s.init(host_block, tail, irs2, max_len=frames)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.05:
            s.process_device(xl, yl, sync=False, order=False)
            s.sync()
        reps = 200
        ts = time.perf_counter()
        for _ in range(reps):
            s.process_device(xl, yl, sync=False, order=False)
        s.sync()
        te = time.perf_counter() - ts
        out[key] = {"value": round(2 * frames * reps / te / 1e6, 1), "unit": "Msamples/s", "ms_per_call": round(te / reps * 1e3, 4),
                    "note": ("one process() call over 40 s of stereo audio, device-resident; " +
                             ("RVC_FLAG_FIXED_PARTITIONS: head 512 + tail 8192 for the whole call" if fixed else
                              "adaptive partitioning: one uniform delay line at block 16384 (P = %d), head / tail stages skipped"
                              % s.partitions(2)))}
        s.close()
    return out


if __name__ == "__main__":
    main()
