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
HBM-bound in; the per-launch fixed cost (4 us launch + a 3 us dependent chain per 512-frame block) is
amortised over 4096 channels: 512 / 1024 / 2048 / 4096 / 8192 channels were measured at
9.2 / 10.9 / 11.5 / 12.5 / 12.2 Gsamples/s (profiles/r2_lockstep/channels_sweep.txt).

Two schedules of that loop are measured in the same run, both strictly causal (nothing of a block
is used before the block has arrived) and both with the reference's partition sizes:
  * `value`: the engine's default, causal TIME TILING -- every 8th block a sweep reads a stage's IR
    spectra and delay line once and leaves partial sums for the next 8 blocks, the blocks in between
    add their few recent partitions: ~3.5x fewer HBM bytes than the reference's loop nest;
  * `reference_schedule`: RVC_FLAG_NO_TIME_TILING, every block sweeps every partition like
    FFTConvolver.cpp:176-187: physical bytes = SURVEY.md 8d's algorithmic bytes, frac <= 1.

A "step" is `--blocks-per-step` consecutive host blocks (default 256 = 16 tail periods = 131072
frames = 2.7 s of audio per channel): 256 per-block launches + 16 tail jobs, so every step does the
same work. Inputs / outputs are resident in HBM (two batches, rotated).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

N > 1: instances are dealt to ranks (unit mod world), every rank runs the same
number of channels (weak scaling), no data-path collective. The gather of output blocks north_star names is ONE RCCL
all_gather per step, issued asynchronously and overlapped with the next step: configs 4 and 5 gather every channel (one
job whose outputs are collected); config 2 hosts thousands of independent instances per GPU whose outputs have no consumer
on the other GPUs, so by default it gathers the output blocks of 8 stereo instances per GPU -- config 4's size -- and
`--gather 2` gathers every channel (at 4096 channels per GPU that is ~400 GB/s inbound per GPU on 8 GPUs, the xGMI
links' whole capacity); `--gather 0` turns it off. One JSON line on rank 0.

Other BASELINE configurations: `--config 4` (8 stereo instances sharded over the ranks, block-synchronous,
strong scaling) and `--config 5` (64 mono channels, 5 s IR, block 4096, offline render = one long
call per step, sharded 64/N per rank, strong scaling).

Side numbers on the same line (rank 0, N = 1): the same loop with 1024 channels per launch, one stereo pair block-synchronously (the plug-in's own
case: latency per block), the offline long-call rate of a stereo pair (adaptive partitioning), the
same forced through the reference's partition sizes, and the CPU baseline.
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


def alg_bytes_block(B: int, P: int) -> int:
    """algorithmic bytes of one block of one reference sub-convolver."""
    return 16 * P * (B + 1) + 8 * (B + 1) + 16 * B


def ref_partitions(head: int, tail: int, ir_len: int):
    """Partition counts of the reference's head / tail0 / tail sub-convolvers (TwoStageFFTConvolver.cpp:117-138)."""
    p_head = -(-min(ir_len, tail) // head)
    p_t0 = -(-min(max(ir_len - tail, 0), tail) // head)
    p_t = -(-max(ir_len - 2 * tail, 0) // tail)
    return p_head, p_t0, p_t


def alg_bytes_per_sample(head: int, tail: int, ir_len: int) -> float:
    """Per channel-sample, reference structure head / tail0 / tail."""
    p_head, p_t0, p_t = ref_partitions(head, tail, ir_len)
    per_tail_block = (tail // head) * (alg_bytes_block(head, p_head) + (alg_bytes_block(head, p_t0) if p_t0 else 0))
    per_tail_block += alg_bytes_block(tail, p_t) if p_t else 0
    return per_tail_block / tail


def flops_per_sample(head: int, tail: int, ir_len: int) -> float:
    """SURVEY.md 8d flop model: 2 * 2.5 N log2 N + 8 P (B+1) per block of each sub-convolver."""
    def blk(B, P):
        N = 2 * B
        return 2 * 2.5 * N * np.log2(N) + 8 * P * (B + 1)
    p_head, p_t0, p_t = ref_partitions(head, tail, ir_len)
    per = (tail // head) * (blk(head, p_head) + (blk(head, p_t0) if p_t0 else 0)) + (blk(tail, p_t) if p_t else 0)
    return float(per / tail)


def make_irs(ir_len: int, instances):
    """One synthetic stereo IR per instance (reevr_amd.synth), generated in parallel."""
    from reevr_amd import synth
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        sets = list(ex.map(lambda i: synth.synth_ir(ir_len, 2, inst=i), instances))
    return [s[c] for s in sets for c in range(2)]


def cpu_baseline(irs, x, head_block: int, budget_s: float) -> dict:
    """The reference itself (oracle/_ref, kind "reference") or the C restatement (kind "port") on this
    host's cores: 512-frame process() calls back to back, tail inline, one instance per thread, driven
    by a pthread loop in C (oracle/cpu_bench.c -- no Python in the loop). Bounded to ~budget_s."""
    from oracle import oracle_py as O
    which = "ref" if O.have_ref() else "orc"
    cores = os.cpu_count() or 1
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    n1, w1, _ = O.cpu_bench(which, 1, head_block, 8192, head_block, irs[:2], x[:2], budget_s * 0.4)
    nm, wm, per = O.cpu_bench(which, cores, head_block, 8192, head_block, irs, x, budget_s * 0.6)
    return {
        "value": round(n1 / w1 / 1e6, 3), "unit": "Msamples/s", "cores": 1,
        "kind": "reference" if which == "ref" else "port",
        "sample": f"mono instance, 10 s IR, head {head_block} / tail 8192, {head_block}-frame process() calls back to back, "
                  f"tail inline: {n1} samples in {w1:.1f} s on 1 thread (C loop, oracle/cpu_bench.c)",
        "cpu_model": model, "flags": "g++ -O3 (SSE2 MAC as shipped, Utilities.cpp:62-111)",
        "all_cores": {"value": round(nm / wm / 1e6, 3), "cores": cores,
                      "sample": f"{cores} independent mono instances ({len(irs)} distinct IRs), one pthread each, "
                                f"{nm} samples in {wm:.1f} s; slowest / fastest thread {min(per)} / {max(per)} samples"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", type=int, default=2, choices=(2, 4, 5), help="BASELINE.json configuration (1-based index)")
    ap.add_argument("--channels", type=int, default=4096, help="config 2: lock-step channels per GPU (2 per stereo instance)")
    ap.add_argument("--time-tiling", type=int, default=1, help="0: RVC_FLAG_NO_TIME_TILING (the reference's per-block sweep order)")
    ap.add_argument("--blocks-per-step", type=int, default=256, help="block-synchronous configs: host blocks per step")
    ap.add_argument("--bg-stream", type=int, default=0, help="1: tail stage on the second HIP stream")
    ap.add_argument("--gather", type=int, default=1,
                    help="N > 1: RCCL all_gather per step; 1: configs 4/5 every channel, config 2 the outputs of 8 stereo "
                         "instances per GPU; 2: every channel; 0: off")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--side", type=int, default=1, help="0: skip the side measurements")
    ap.add_argument("--watchdog", type=float, default=1500.0,
                    help="seconds after which a stuck run dumps every thread's stack and exits (0: off)")
    args = ap.parse_args()
    if args.watchdog > 0:          # a hung collective or kernel must not hold the GPU box until the driver's limit
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

    import torch
    import reevr_amd
    from reevr_amd import KERNEL_NAMES, shard, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...`")
    # development switch: all ranks on GPU 0 over gloo, to exercise the N > 1 control flow on a 1-GPU box
    same_device = os.environ.get("REEVR_BENCH_SAME_DEVICE") == "1"
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
    if args.config == 2:
        ir_len, host_block, long_call = 10 * SR, 512, False
        if args.channels < 2 or args.channels % 2:
            raise SystemExit("--channels must be a positive even number (stereo instances)")
        n_inst = args.channels // 2
        instances = shard.units_for_rank(n_inst * world, world, rank)  # rank = unit mod world, equal shards
        scaling = "weak"
        workload = (f"stereo, 10 s IR @ 48 kHz, block=512 (head 512 / tail 8192), TwoStage convolver: {n_inst} stereo "
                    f"instances per GPU in lock-step, one process() per 512-frame block")
    elif args.config == 4:
        ir_len, host_block, long_call = 10 * SR, 512, False
        if 8 % world:
            raise SystemExit("--config 4 shards 8 stereo instances: --gpus must divide 8")
        instances = shard.units_for_rank(8, world, rank)
        scaling = "strong"
        workload = "8 independent stereo instances, 10 s IR @ 48 kHz, block=512, sharded over the GPUs (unit mod world)"
    else:
        ir_len, host_block, long_call = 5 * SR, 4096, True
        if 32 % world:
            raise SystemExit("--config 5 shards 64 mono channels (32 pairs): --gpus must divide 32")
        instances = shard.units_for_rank(32, world, rank)
        scaling = "strong"
        workload = "batched offline render: 64 mono channels, 5 s IR @ 48 kHz, block=4096, 64/N channels per GPU, one long call per step"
    nch = 2 * len(instances)
    head = 1
    while head < host_block:
        head *= 2
    tail = max(8192, 2 * head)                                        # StereoConvolver.cpp:11-15
    if not long_call and (args.blocks_per_step < 1 or (args.blocks_per_step * host_block) % tail):
        raise SystemExit("--blocks-per-step must cover whole tail periods (a multiple of %d)" % (tail // host_block))
    frames_step = 20 * SR if long_call else args.blocks_per_step * host_block   # frames per channel per step
    frames_step -= frames_step % host_block
    nbuf = 1 if long_call else 2                                      # input / output batches rotated through

    t_gen = time.perf_counter()
    irs = make_irs(ir_len, instances)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        x = np.stack(list(ex.map(lambda uc: synth.synth_input(frames_step * nbuf, 2 * uc[0] + uc[1]),
                                 [(u, c) for u in instances for c in range(2)])))
    gen_s = time.perf_counter() - t_gen
    conv = reevr_amd.ConvolverSet(nch, device=local_rank, bg_stream=bool(args.bg_stream), time_tiling=bool(args.time_tiling))
    t_init = time.perf_counter()
    if not conv.init(host_block, tail, irs, max_len=frames_step if long_call else host_block):
        raise SystemExit(f"init failed: {conv.last_error_string}")
    conv.sync()
    init_ms = (time.perf_counter() - t_init) * 1e3
    d_in = torch.from_numpy(x).to(dev)
    d_out = torch.empty_like(d_in)
    do_gather = bool(args.gather and world > 1)
    torch.cuda.synchronize()
    state = {"i": 0}
    # N > 1: the output batch of every step is gathered with ONE all_gather, overlapped with the next step's compute:
    # the collective is ordered behind this step's kernels (torch's current stream waits for the set's stream) and runs
    # on the communicator's stream; a batch buffer is reused only after its own gather has completed.
    gch = nch if (args.gather >= 2 or args.config != 2) else min(nch, 16)      # channels per rank that are gathered
    g_out = [torch.empty((world,) + (gch, frames_step), dtype=torch.float32, device=dev) for _ in range(nbuf)] if do_gather else None
    g_stage = [torch.empty((gch, frames_step), dtype=torch.float32, device=dev) for _ in range(nbuf)] if do_gather else None
    pending = [None] * nbuf

    def step(gather=True):
        b = state["i"] % nbuf
        state["i"] += 1
        xi = d_in[:, b * frames_step:(b + 1) * frames_step]
        if do_gather and gather:
            if pending[b] is not None:         # the gather that read this batch's buffers two steps ago
                pending[b].wait()
                pending[b] = None
            # (every channel gathered: the set writes the contiguous staging batch the collective reads in place)
            yo = g_stage[b] if gch == nch else d_out[:, b * frames_step:(b + 1) * frames_step]
            if long_call:
                conv.process_device(xi, yo, sync=False, order=True)
            else:
                conv.process_device_blocks(xi, host_block, yo, sync=False, order=True)
            if gch != nch:
                g_stage[b].copy_(yo[:gch])     # (torch's current stream, ordered behind the set's stream by order=True)
                yo = g_stage[b]
            pending[b] = shard.gather_batches_async(yo, g_out[b], dist)
            return
        yo = d_out[:, b * frames_step:(b + 1) * frames_step]
        if long_call:
            conv.process_device(xi, yo, sync=False, order=False)
        else:                                  # the host's per-block loop (in C): one call per 512-frame block
            conv.process_device_blocks(xi, host_block, yo, sync=False, order=False)

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

    # untimed pre-roll to the steady state: the tail delay line holds P_T tail blocks of history and
    # rows before time 0 are never fetched, so the first P_T + 2 tail periods move fewer bytes
    pre = 0
    if not long_call:
        pre = -(-(conv.partitions(1) + 4) * tail // frames_step)
        for _ in range(pre):
            step()
        conv.sync()
    else:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.05:
            step()
            conv.sync()
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
    elapsed = shard.max_over_ranks(elapsed, dist, dev)    # slowest rank
    total_ch = nch * world if args.config == 2 else (16 if args.config == 4 else 64)
    total_samples = total_ch * frames_step * args.steps
    value = total_samples / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- per-kernel durations, live, with HIP events on the stream the kernels run on ----
    conv.set_timing(True)
    conv.kernel_time_reset()
    ksteps = 1
    for _ in range(ksteps):
        step(gather=False)                     # (rank 0 only: no collective here)
    conv.sync()
    kern = {}
    for kid, name in enumerate(KERNEL_NAMES):
        n, ms = conv.kernel_time(kid)
        if n:
            kern[name] = {"launches_per_step": n / ksteps, "avg_ms": ms / n}
    conv.set_timing(False)
    conv.kernel_time_reset()
    PA, PT = conv.partitions(0), conv.partitions(1)
    p_head, p_t0, p_t = ref_partitions(head, tail, ir_len)
    tiled = bool(args.time_tiling) and "sweep_tail" in kern
    K = 8                                                # rvc::kSweepRows
    row_h, row_t = 8.0 * head * nch, 8.0 * tail * nch    # bytes of one spectrum row of every channel
    io_blk = nch * (4.0 * 3 * head + 4.0 * 2 * head)     # per block: input + history + tail ring read, output + ring written
    # Bytes per LAUNCH of each kernel family. Reference schedule: SURVEY.md 8d's algorithmic figures (what the
    # reference's loop nest moves = what these kernels move). Time-tiled schedule: the bytes of the structure
    # actually executed, so that frac <= 1 means what it says.
    if tiled:
        exe = {
            # audio part (H0, H1, X_{k-1}, accumulator read; X_k written; samples) + on average (K-1)/2 recent partitions patched
            "fused_block": 5 * row_h + io_blk + ((K - 1) / 2.0 * 2 + 2) * row_h * (K - 1) / K,
            "sweep_head": (PA + (PA - 2) + K) * row_h,   # IR rows + arrived delay-line rows read once, K partial rows written
            "sweep_tail": (2 * PT + K) * row_t,
            "fir_tail": ((K / 2.0) * 2 + 2) * row_t,      # patch: t = 1..K-1 recent partitions (mean K/2) + the sweep row, 1 row out
            "premultiply": 2.0 * (PA - 2) * row_h + row_h,
            "fft_fwd_tail": float(nch * (4 * 2 * tail + 8 * tail)),
            "fft_inv_tail": float(nch * (8 * tail + 4 * tail)),
        }
    else:
        exe = {
            # the per-block launch does one head block of the reference's head AND tail0 sub-convolvers
            "fused_block": float(nch * (alg_bytes_block(head, p_head) + (alg_bytes_block(head, p_t0) if p_t0 else 0))),
            "premultiply": float(nch * 16 * PA * (head + 1)),
            "fir_head": float(nch * 16 * PA * (head + 1)),
            "fir_tail": float(nch * 16 * p_t * (tail + 1)),
            "fft_fwd_head": float(nch * (4 * head + 8 * (head + 1))),
            "fft_inv_head": float(nch * (8 * (head + 1) + 12 * head)),
            "fft_fwd_tail": float(nch * (4 * tail + 8 * (tail + 1))),
            "fft_inv_tail": float(nch * (8 * (tail + 1) + 12 * tail)),
            "ingest": float(nch * 8 * host_block),
        }
    traffic_all, tsrc = {}, None
    tpath = os.path.join(ROOT, "profiles", "r2_traffic.json")
    if os.path.exists(tpath) and not long_call:
        tj = json.load(open(tpath))
        if tj.get("channels") == nch and tj.get("config") == args.config and bool(tj.get("time_tiling", 0)) == tiled:
            traffic_all = {k: v["traffic_bytes"] for k, v in tj.get("kernels", {}).items()}
            tsrc = "profiles/r2_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950-corrected)"
    roof_all = {}
    exe_bytes_step = 0.0
    for k, v in kern.items():
        if long_call or k not in exe:
            continue
        gbs = exe[k] / (v["avg_ms"] * 1e-3) / 1e9
        exe_bytes_step += exe[k] * v["launches_per_step"]
        roof_all[k] = {"launches_per_step": v["launches_per_step"], "avg_launch_ms": round(v["avg_ms"], 5),
                       "ms_per_step": round(v["avg_ms"] * v["launches_per_step"], 5),
                       "bytes_per_launch": exe[k], "achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                       "traffic": traffic_all.get(k)}
    roof = None
    if roof_all:
        dominant = max(roof_all, key=lambda k: roof_all[k]["ms_per_step"])      # largest share of a step
        r = roof_all[dominant]
        roof = {"bound": "hbm", "kernel": dominant, "achieved": r["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": r["frac"], "traffic": r["traffic"], "bytes_per_launch": r["bytes_per_launch"],
                "avg_launch_ms": r["avg_launch_ms"], "traffic_source": tsrc,
                "note": ("time-tiled block-synchronous schedule: bytes_per_launch = the bytes of the structure this launch executes "
                         "(a sweep reads the stage's IR spectra and arrived delay-line rows ONCE per 8 blocks; a patch / the per-block "
                         "launch only the partitions that arrived since), averaged over the launches of its family; frac <= 1"
                         if tiled else
                         "reference schedule: every launch re-reads the IR spectra and the delay line of its stage once (16 B per "
                         "partition x bin), physical HBM bytes = SURVEY.md 8d algorithmic bytes, frac <= 1") +
                        "; 0.5-4 flop/B, far below the fp32 ridge"}
    bps = alg_bytes_per_sample(head, tail, ir_len)
    fps = flops_per_sample(head, tail, ir_len)
    rate_gpu = value / world * 1e6
    path = {"executed_bytes_per_sample": round(exe_bytes_step / (nch * frames_step), 1) if exe_bytes_step else None,
            "executed_GBs_per_gpu": round(rate_gpu * exe_bytes_step / (nch * frames_step) / 1e9, 1) if exe_bytes_step else None,
            "frac_of_hbm_peak": round(rate_gpu * exe_bytes_step / (nch * frames_step) / 1e9 / HBM_PEAK_GBS, 4) if exe_bytes_step else None,
            "reference_alg_bytes_per_sample": round(bps, 1),
            "reference_alg_GBs_equivalent": round(rate_gpu * bps / 1e9, 1),
            "reference_alg_frac_equivalent": round(rate_gpu * bps / 1e9 / HBM_PEAK_GBS, 4),
            "flops_per_sample": round(fps, 1), "frac_of_fp32_peak": round(rate_gpu * fps / 1e12 / FP32_PEAK_TFLOPS, 4),
            "x_realtime_per_gpu": round(rate_gpu / SR, 1),
            "note": "frac_of_hbm_peak = bytes the executed schedule moves per channel-sample x measured rate / 8 TB/s (<= 1). "
                    "reference_alg_*: SURVEY.md 8d's 1497 B/sample of the REFERENCE's loop nest x the same rate -- with time tiling "
                    "this exceeds the physical traffic by the tiling's byte saving and is a throughput equivalent, not an "
                    "efficiency; the `reference_schedule` entry is the run where the two coincide."}

    side = {}
    if args.side and args.config == 2 and world == 1:
        conv.close()
        if tiled:       # the same loop in the reference's sweep order (same channels, same inputs)
            rconv = reevr_amd.ConvolverSet(nch, device=local_rank, bg_stream=bool(args.bg_stream), time_tiling=False)
            assert rconv.init(host_block, tail, irs, max_len=host_block)

            def rstep(i):
                b = i % nbuf
                rconv.process_device_blocks(d_in[:, b * frames_step:(b + 1) * frames_step], host_block,
                                            d_out[:, b * frames_step:(b + 1) * frames_step], sync=False, order=False)
            for i in range(pre + 1):
                rstep(i)
            rconv.sync()
            rsteps = max(2, args.steps // 4)
            tr = time.perf_counter()
            for i in range(rsteps):
                rstep(i)
            rconv.sync()
            tr = time.perf_counter() - tr
            rrate = nch * frames_step * rsteps / tr
            side["reference_schedule"] = {
                "value": round(rrate / 1e6, 3), "unit": "Msamples/s", "steps": rsteps, "ms_per_step": round(tr / rsteps * 1e3, 4),
                "alg_bytes_per_sample": round(bps, 1), "achieved_GBs": round(rrate * bps / 1e9, 1),
                "frac_of_hbm_peak": round(rrate * bps / 1e9 / HBM_PEAK_GBS, 4),
                "note": "RVC_FLAG_NO_TIME_TILING: same channels / inputs / call pattern, every 512-frame block sweeps all 32 + 57 "
                        "partitions (FFTConvolver.cpp:176-187): physical bytes = SURVEY.md 8d algorithmic bytes"}
            rconv.close()
        if tiled and nch > 1024:   # the same loop with a quarter of the channels per launch (the fixed cost of a launch shows)
            qn = 1024
            qconv = reevr_amd.ConvolverSet(qn, device=local_rank, bg_stream=bool(args.bg_stream), time_tiling=True)
            assert qconv.init(host_block, tail, irs[:qn], max_len=host_block)

            def qstep(i):
                b = i % nbuf
                qconv.process_device_blocks(d_in[:qn, b * frames_step:(b + 1) * frames_step], host_block,
                                            d_out[:qn, b * frames_step:(b + 1) * frames_step], sync=False, order=False)
            for i in range(pre + 1):
                qstep(i)
            qconv.sync()
            qsteps = max(4, args.steps // 2)
            tq = time.perf_counter()
            for i in range(qsteps):
                qstep(i)
            qconv.sync()
            tq = time.perf_counter() - tq
            side["lockstep_1024_channels"] = {"value": round(qn * frames_step * qsteps / tq / 1e6, 3), "unit": "Msamples/s",
                                              "steps": qsteps, "ms_per_step": round(tq / qsteps * 1e3, 4),
                                              "note": "same loop, schedule and kernels with 1024 channels (512 stereo instances, "
                                                      "8 GB resident) per launch"}
            qconv.close()
        del d_in, d_out
        torch.cuda.empty_cache()
        side.update(side_measurements(torch, reevr_amd, synth, irs[:2], local_rank, dev, host_block, tail))
    cpu = None
    if world == 1 and args.cpu_seconds > 0 and args.config == 2:
        cores = os.cpu_count() or 1
        n_cpu_irs = min(len(irs), max(2, cores))
        xin = [np.ascontiguousarray(x[c % nch, :frames_step * nbuf]) for c in range(min(n_cpu_irs, nch))]
        cpu = cpu_baseline(irs[:n_cpu_irs], xin, host_block, args.cpu_seconds)

    line = {
        "metric": "Msamples/s convolved (stereo, 10s IR, block=512); % HBM roofline",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "baseline_config": args.config,
                   "channels_per_gpu": nch, "stereo_instances_per_gpu": nch // 2, "instances_total": total_ch // 2,
                   "frames_per_channel_per_step": frames_step, "host_block": host_block,
                   "calls_per_step": 1 if long_call else frames_step // host_block,
                   "partitions": {"head+tail0 (block %d)" % head: PA, "tail (block %d)" % tail: PT},
                   "resident_GB": round(nch * 8.0 * (PA * head + (PT + 2) * tail) * 2 / 1e9, 2),
                   "schedule": "long call" if long_call else ("causal time tiling (8-block sweeps + patches)" if tiled
                                                                else "reference order (RVC_FLAG_NO_TIME_TILING)"),
                   "call": ("one process() per step" if long_call else
                            "one process_device() per 512-frame host block for all channels (rvc_set_process_device_blocks), "
                            "device-resident I/O, %d input/output batches rotated" % nbuf),
                   "pre_roll_steps": pre, "gather": do_gather, "gathered_channels_per_gpu": gch if do_gather else 0,
                   "sharding": "instances dealt to ranks, equal shards, no data-path collective"
                               + (f"; one RCCL all_gather of the output blocks of {gch} channels per GPU per step, overlapped with the next step" if do_gather else "")},
        "roofline": roof,
        "roofline_all": roof_all,
        "path_roofline": path,
        "kernels_ms": {k: round(v["avg_ms"], 5) for k, v in kern.items()},
        **side,
        "cpu_baseline": cpu,
        "init_ms": round(init_ms, 2), "synth_s": round(gen_s, 2),
    }
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def side_measurements(torch, reevr_amd, synth, irs2, local_rank, dev, host_block, tail):
    """ONE stereo pair (the plug-in's own case), three ways. Not the headline."""
    out = {}
    # (a) block-synchronous: per 512-frame block, device-resident loop and host-pointer calls (the audio thread's view),
    #     with one launch per block (tail job inline / on the second stream) and with the resident kernel
    nblk = 3000
    xs_h = np.stack([synth.synth_input(host_block * nblk, c) for c in range(2)])
    xs = torch.from_numpy(xs_h).to(dev)
    ys = torch.empty_like(xs)
    torch.cuda.synchronize()
    res = {}
    for mode, kw in (("tail_on_second_stream", dict(bg_stream=True)), ("tail_inline", dict(bg_stream=False)),
                     ("persistent_kernel", dict(bg_stream=True, persistent=True))):
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
                                        "(pinned staging + hand-off + kernel + copy back; stopwatch in C). persistent_kernel = "
                                        "RVC_FLAG_PERSISTENT (resident kernel fed through a doorbell, no launch per block)"}
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
