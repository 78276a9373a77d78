#!/usr/bin/env python3
"""bench.py -- throughput of the partitioned-convolution hot path on MI355X.

Workload (BASELINE.json configs[1]): stereo, 10 s IR @ 48 kHz, host block 512
(-> head 512 / tail 8192, StereoConvolver.cpp:11-15), synthetic white-noise input,
decaying-noise IR with the reference's auto-gain (reevr_amd/synth.py).

A "step" is one pass of the hot path over one batch of input: ONE process() call of
`--frames` frames per channel (default 40 s = 1 920 000 frames = 3750 blocks of 512) with
input and output resident in HBM. process() takes any length, and its result does not
depend on how the stream is cut into calls (tests/test_gpu_parity.py), so this is the same
function the plugin calls per 512-sample block -- batched in time because a single
512-frame block (2 KB per channel) cannot fill a 256-CU GPU. Two more numbers are reported
beside `value` so nothing hides behind the batching:
  * "two_stage": the same call with RVC_FLAG_FIXED_PARTITIONS, i.e. forced through the
    reference's head-512 / tail-8192 partition structure (the engine's default instead gives a
    call that spans >= 4 tail blocks to one uniform delay line at the tail block size -- same
    output, no 512-sample work where no 512-sample latency is asked for);
  * "streaming": strictly one call per 512-frame block (the plugin's real-time pattern).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

N > 1: every rank owns one independent stereo instance (its own IR, `inst = rank`), no
data-path collective (weak scaling; `--gather` adds the optional RCCL all_gather of the
output batch). One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SR = 48000
IR_LEN = 10 * SR
HOST_BLOCK = 512
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


ROOF_NOTES = {
    "fir": "achieved = ALGORITHMIC bytes of the reference structure this launch replaces (16 B per partition x bin of "
           "the head, tail0 and tail delay lines, SURVEY.md 8d) / measured launch time. The kernel tiles 64 blocks of "
           "time per workgroup and, for long calls, uses one delay line at a larger block for the whole IR, so physical "
           "HBM traffic (`traffic`) is ~50x below the algorithmic figure: frac > 1 is expected, the kernel is fp32-FMA "
           "bound (DESIGN.md 7).",
    "fft": "achieved = ALGORITHMIC bytes of the reference transforms this launch replaces (4 B per input sample + 8 B "
           "per spectrum bin, for every head AND tail block of the call, SURVEY.md 8d) / measured launch time; `traffic` "
           "is what the kernel physically moved (overlap-save reads 2B samples per B-sample block, one transform per "
           "16384-sample block on the long-call path). The FIR kernel takes about the same time per step "
           "(kernels_ms / roofline_all); which of the two is 'dominant' can flip run to run.",
}


def alg_bytes_block(B: int, P: int) -> int:
    """algorithmic bytes of one block of one reference sub-convolver."""
    return 16 * P * (B + 1) + 8 * (B + 1) + 16 * B


def alg_bytes_per_sample(head: int, tail: int, ir_len: int) -> float:
    """Per channel-sample, reference structure head / tail0 / tail (TwoStageFFTConvolver.cpp:117-138)."""
    p_head = -(-min(ir_len, tail) // head)
    p_t0 = -(-min(max(ir_len - tail, 0), tail) // head)
    p_t = -(-max(ir_len - 2 * tail, 0) // tail)
    per_tail_block = (tail // head) * (alg_bytes_block(head, p_head) + (alg_bytes_block(head, p_t0) if p_t0 else 0))
    per_tail_block += alg_bytes_block(tail, p_t) if p_t else 0
    return per_tail_block / tail


def cpu_baseline(irs: np.ndarray, x: np.ndarray, budget_s: float) -> dict:
    """The reference itself (oracle/_ref, kind "reference") or the C restatement (kind "port")
    on this host's cores, 512-frame process() calls, tail inline, bounded to ~budget_s."""
    from oracle import oracle_py as O
    which = "ref" if O.have_ref() else "orc"
    nch, frames = x.shape
    frames -= frames % HOST_BLOCK

    def run_channel(c: int, stop_at: float, counter: list):
        conv = O.TwoStageFFTConvolver(which)
        assert conv.init(HOST_BLOCK, 8192, irs[c % irs.shape[0]])
        fn = conv._b.fn("twostage_process")
        xin = np.ascontiguousarray(x[c % nch])
        out = np.empty(HOST_BLOCK, np.float32)
        import ctypes as C
        fp = C.POINTER(C.c_float)
        outp = out.ctypes.data_as(fp)
        base = xin.ctypes.data
        done = 0
        while True:
            for i in range(0, frames, HOST_BLOCK):
                fn(conv._h, C.cast(base + 4 * i, fp), outp, HOST_BLOCK)
            done += frames
            if time.perf_counter() >= stop_at:
                break
        counter[c] = done

    # (i) one thread, channels one after the other -- comparable with BASELINE.md section 2
    t0 = time.perf_counter()
    cnt = [0] * nch
    per = budget_s / (2 * nch)
    for c in range(nch):
        run_channel(c, time.perf_counter() + per, cnt)
    t1 = time.perf_counter()
    single = sum(cnt) / (t1 - t0) / 1e6
    # (ii) all host cores, one convolver instance per thread (ctypes releases the GIL)
    cores = os.cpu_count() or 1
    cntm = [0] * cores
    t2 = time.perf_counter()
    stop = t2 + budget_s / 2
    th = [threading.Thread(target=run_channel, args=(c, stop, cntm)) for c in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    t3 = time.perf_counter()
    multi = sum(cntm) / (t3 - t2) / 1e6
    return {
        "value": round(single, 3), "unit": "Msamples/s", "cores": 1,
        "kind": "reference" if which == "ref" else "port",
        "sample": f"stereo 10 s IR, head 512 / tail 8192, {HOST_BLOCK}-frame process() calls, tail inline, "
                  f"{sum(cnt)} channel-samples in {t1 - t0:.1f} s on 1 thread",
        "all_cores": {"value": round(multi, 3), "cores": cores,
                      "sample": f"{cores} independent mono instances, one per thread, {t3 - t2:.1f} s"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=40 * SR, help="frames per channel per step (one process() call)")
    ap.add_argument("--bg-stream", type=int, default=0, help="1: tail stage on the second HIP stream (overlaps the head stage)")
    ap.add_argument("--fixed-partitions", type=int, default=0,
                    help="1: force the reference's head/tail partition sizes even for long calls")
    ap.add_argument("--gather", action="store_true", help="RCCL all_gather of the output batch each step")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--side", type=int, default=1, help="0: skip the two_stage side measurement")
    ap.add_argument("--stream-calls", type=int, default=3000, help="512-frame calls of the streaming side measurement")
    args = ap.parse_args()

    import torch
    import reevr_amd
    from reevr_amd import KERNEL_NAMES, shard, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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

    frames = int(args.frames)
    nch = 2
    irs = synth.synth_ir(IR_LEN, nch, inst=rank)                 # this rank's stereo instance
    x = np.stack([synth.synth_input(frames, c + 2 * rank) for c in range(nch)])
    conv = reevr_amd.ConvolverSet(nch, device=local_rank, bg_stream=bool(args.bg_stream),
                                  fixed_partitions=bool(args.fixed_partitions))
    t_init = time.perf_counter()
    if not conv.init(HOST_BLOCK, 8192, list(irs), max_len=frames):
        raise SystemExit(f"init failed: {conv.last_error_string}")
    conv.sync()
    init_ms = (time.perf_counter() - t_init) * 1e3
    head, tail = conv.head_block, conv.tail_block
    d_in = torch.from_numpy(x).to(dev)
    d_out = torch.empty_like(d_in)
    do_gather = bool(args.gather and world > 1)
    torch.cuda.synchronize()

    def step():
        conv.process_device(d_in, d_out, sync=False, order=False)   # buffers resident and complete; synced below
        if do_gather:                      # one RCCL all_gather per batch of 3750 blocks
            conv.sync()
            shard.gather_batches(d_out, dist)

    def fence():
        conv.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # untimed pre-roll: ~50 ms of the same step so that the GPU's clocks have settled whatever K / W are
    # (a step is 60 us: with K = 50 the whole timed region would otherwise end before they have)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.05:
        for _ in range(20):
            step()
        conv.sync()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    # closing bracket: synchronise this rank, stamp, barrier; the reported time is the MAX over ranks of
    # the stamped spans (all ranks left the opening barrier together), so the collective's own latency
    # (~0.2 ms for an RCCL barrier, 3 steps' worth) is not charged to the K steps
    conv.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    conv.check()
    elapsed = shard.max_over_ranks(elapsed, dist, dev)    # slowest rank
    total_samples = world * nch * frames * args.steps
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
    ksteps = min(args.steps, 10)
    for _ in range(ksteps):
        conv.process_device(d_in, d_out, sync=True)
    kern = {}
    for kid, name in enumerate(KERNEL_NAMES):
        n, ms = conv.kernel_time(kid)
        if n:
            kern[name] = {"launches": n, "avg_ms": ms / n}
    conv.set_timing(False)
    conv.kernel_time_reset()
    PA, PT = conv.partitions(0), conv.partitions(1)
    rows_A = -(-frames // head) + (1 if frames % head else 0)
    # algorithmic bytes per launch of each kernel family:
    #   FIR:  16 B per (partition, bin) pair  = read one IR bin + one delay-line bin (8 B each)
    blocksA, blocksT = frames / head, frames / tail
    alg = {
        "fir_head": 16.0 * PA * (head + 1) * blocksA * nch,
        "fir_tail": 16.0 * PT * (tail + 1) * blocksT * nch,
        "fft_fwd_head": (4.0 * head + 8.0 * (head + 1)) * blocksA * nch,
        "fft_inv_head": (8.0 * (head + 1) + 12.0 * head) * blocksA * nch,
        "fft_fwd_tail": (4.0 * tail + 8.0 * (tail + 1)) * blocksT * nch,
        "fft_inv_tail": (8.0 * (tail + 1) + 12.0 * tail) * blocksT * nch,
        "ingest": 8.0 * frames * nch,
    }
    adaptive = not args.fixed_partitions and not args.bg_stream and "fir_head" not in kern
    if adaptive:
        # the long-call path runs ONE delay line (P_T + 2 partitions at the tail block size) that does
        # the work of the reference's head, tail0 and tail delay lines: it inherits all their
        # algorithmic bytes; likewise the transforms
        alg["fir_tail"] += alg["fir_head"]
        alg["fft_fwd_tail"] += alg["fft_fwd_head"]
        alg["fft_inv_tail"] += alg["fft_inv_head"]
    dominant = max(kern, key=lambda k: kern[k]["avg_ms"]) if kern else None
    roof = None
    if dominant:
        ach = alg[dominant] / (kern[dominant]["avg_ms"] * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters cannot be read from inside this process:
        # they come from the committed rocprofv3 --pmc passes of this same command
        # (profiles/traffic.json, made by tools/pmc_summarize.py; valid for the default --frames only).
        traffic, tsrc = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and frames == 40 * SR:
            tj = json.load(open(tpath))
            if dominant in tj.get("kernels", {}):
                traffic = tj["kernels"][dominant]["traffic_bytes"]
                tsrc = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950-corrected)"
        roof = {"bound": "hbm", "kernel": dominant, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                "alg_bytes_per_launch": alg[dominant], "avg_launch_ms": round(kern[dominant]["avg_ms"], 5),
                "traffic_source": tsrc,
                "note": ROOF_NOTES["fir" if dominant.startswith("fir") else "fft"]}
    # every timed kernel family against the HBM roofline (algorithmic numerator), for the record
    roof_all = {k: {"avg_launch_ms": round(v["avg_ms"], 5),
                    "achieved_GBs": round(alg[k] / (v["avg_ms"] * 1e-3) / 1e9, 1),
                    "frac": round(alg[k] / (v["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for k, v in kern.items()}
    # and the PHYSICAL rate: measured HBM bytes per launch (PMC passes, profiles/traffic.json) / launch time
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and frames == 40 * SR:
        tk = json.load(open(tpath)).get("kernels", {})
        for k, v in kern.items():
            if k in tk:
                gbs = tk[k]["traffic_bytes"] / (v["avg_ms"] * 1e-3) / 1e9
                roof_all[k]["traffic"] = tk[k]["traffic_bytes"]
                roof_all[k]["physical_GBs"] = round(gbs, 1)
                roof_all[k]["physical_frac"] = round(gbs / HBM_PEAK_GBS, 4)
    bps = alg_bytes_per_sample(head, tail, IR_LEN)
    path_gbs = value / world * 1e6 * bps / 1e9

    # ---- side measurement: the same call forced through the reference's partition sizes ----
    two_stage = None
    if adaptive and args.side:
        fconv = reevr_amd.ConvolverSet(nch, device=local_rank, fixed_partitions=True)
        assert fconv.init(HOST_BLOCK, 8192, list(irs), max_len=frames)
        for _ in range(max(args.warmup, 2)):
            fconv.process_device(d_in, d_out, sync=False, order=False)
        fconv.sync()
        tf = time.perf_counter()
        for _ in range(args.steps):
            fconv.process_device(d_in, d_out, sync=False, order=False)
        fconv.sync()
        tf = time.perf_counter() - tf
        two_stage = {"value": round(nch * frames * args.steps / tf / 1e6, 3), "unit": "Msamples/s",
                     "ms_per_step": round(tf / args.steps * 1e3, 4),
                     "note": "RVC_FLAG_FIXED_PARTITIONS: head 512 (32 partitions) + tail 8192 (57 partitions) for the whole call"}
        fconv.close()

    # ---- streaming side measurement: one process() call per 512-frame host block --------
    streaming = None
    if args.stream_calls > 0:
        nblk = args.stream_calls
        s_in = d_in[:, :HOST_BLOCK * nblk].contiguous()
        s_out = torch.empty_like(s_in)
        res = {}
        for mode, bg in (("tail_on_second_stream", True), ("tail_inline", False)):
            sconv = reevr_amd.ConvolverSet(nch, device=local_rank, bg_stream=bg)
            assert sconv.init(HOST_BLOCK, 8192, list(irs), max_len=HOST_BLOCK)
            sconv.process_device_blocks(s_in[:, :HOST_BLOCK * 200].contiguous(), HOST_BLOCK)   # warm-up
            ts = time.perf_counter()
            sconv.process_device_blocks(s_in, HOST_BLOCK, s_out)      # the per-block host loop, in C
            te = time.perf_counter() - ts
            res[mode] = {"Msamples_s": round(nch * HOST_BLOCK * nblk / te / 1e6, 3),
                         "us_per_block": round(te / nblk * 1e6, 2)}
            sconv.close()
        best = max(res.values(), key=lambda r: r["Msamples_s"])
        streaming = {"value": best["Msamples_s"], "unit": "Msamples/s", "us_per_block": best["us_per_block"],
                     "modes": res,
                     "note": "one process_device() call per 512-frame block (host loop in C, "
                             "rvc_set_process_device_blocks): ONE launch per block (fused latency kernel with the "
                             "next block's partial accumulator appended); tail job every 16 blocks on the second "
                             "stream (RVC_FLAG_BG_STREAM, lowest per-call latency) or inline (highest rate)"}

    cpu = cpu_baseline(irs, x[:, :20 * SR], args.cpu_seconds) if (world == 1 and args.cpu_seconds > 0) else None

    line = {
        "metric": "Msamples/s convolved (stereo, 10s IR, block=512); % HBM roofline",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "stereo, 10 s IR @ 48 kHz, block=512 (head 512 / tail 8192), TwoStage convolver",
                   "frames_per_step": frames, "channels_per_gpu": nch, "instances": world,
                   "partitions": {"head+tail0": PA, "tail": PT},
                   "call": "one process() per step, device-resident I/O", "gather": do_gather,
                   "partitioning": "fixed head/tail" if not adaptive else
                                   ("adaptive: long call -> one uniform delay line at block 16384 (P = %d)" % conv.partitions(2)
                                    if conv.partitions(2) > 0 and frames >= 4 * 16384 else
                                    "adaptive: long call -> one uniform delay line at the tail block size (P = %d)" % (PT + 2)),
                   "sharding": "one independent stereo instance per rank (unit mod world), no data-path collective"},
        "roofline": roof,
        "roofline_all": roof_all,
        "path_roofline": {"alg_bytes_per_sample": round(bps, 1), "achieved_GBs_per_gpu": round(path_gbs, 1),
                          "frac_of_hbm_peak": round(path_gbs / HBM_PEAK_GBS, 4),
                          "x_realtime_per_gpu": round(value / world * 1e6 / (SR * nch), 1)},
        "kernels_ms": {k: round(v["avg_ms"], 5) for k, v in kern.items()},
        "two_stage": two_stage,
        "streaming": streaming,
        "cpu_baseline": cpu,
        "init_ms": round(init_ms, 2),
    }
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
