"""Multi-GPU path on real devices. Two tests:

* bench.py's N > 1 control flow (sharding of units, per-step all_gather, max-over-ranks timing, one JSON
  line) for BASELINE configs 2, 4 and 5, run as two ranks under torch.distributed.run. On a 1-GPU box the
  ranks share device 0 over gloo (REEVR_BENCH_SAME_DEVICE=1); with >= 2 devices they take one each over RCCL.
* the RCCL branch of shard.gather_batches with real engines on two devices against the oracle; skipped when
  fewer than two devices are visible.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ndev():
    from reevr_amd import _lib
    return _lib.lib().rvc_device_count()


@pytest.mark.parametrize("extra", [["--config", "2", "--channels", "8", "--blocks-per-step", "32"],
                                   ["--config", "2", "--channels", "40", "--blocks-per-step", "32"],
                                   ["--config", "2", "--channels", "40", "--blocks-per-step", "32", "--gather", "2"],
                                   ["--config", "4", "--blocks-per-step", "32"],
                                   ["--config", "5"]], ids=["cfg2", "cfg2_bounded_gather", "cfg2_full_gather", "cfg4", "cfg5"])
def test_bench_two_ranks(extra):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if _ndev() < 2:
        env["REEVR_BENCH_SAME_DEVICE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--cpu-seconds", "0", "--side", "0", "--watchdog", "120"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["value"] > 0
    assert rec["config"]["gather"] is True
    want_scaling = "weak" if extra[1] == "2" else "strong"
    assert rec["scaling"] == want_scaling
    if extra[1] == "2":
        nch = int(extra[3])
        assert rec["config"]["channels_per_gpu"] == nch and rec["config"]["instances_total"] == nch
        # default: the output blocks of 8 stereo instances per GPU are gathered; --gather 2: every channel
        assert rec["config"]["gathered_channels_per_gpu"] == (nch if "--gather" in extra else min(nch, 16))
    if extra[1] == "4":
        assert rec["config"]["channels_per_gpu"] == 8          # 8 stereo instances over 2 ranks
    if extra[1] == "5":
        assert rec["config"]["channels_per_gpu"] == 32         # 64 mono channels over 2 ranks


@pytest.mark.parametrize("extra", [["--config", "2", "--channels", "4", "--blocks-per-step", "16"],
                                   ["--config", "4", "--blocks-per-step", "16"],
                                   ["--config", "5"]], ids=["cfg2", "cfg4", "cfg5"])
def test_bench_eight_ranks_control_flow(extra):
    """The N = 8 control flow the driver's 8-GPU node will meet, before it meets it: eight ranks under torch.distributed.run
    (one device each over RCCL when eight are visible, else all on device 0 over gloo): units_for_rank 8-way, the
    strong-scaling totals of configs 4 / 5 (one stereo instance / four channel pairs per rank), the gather buffer shapes,
    max-over-ranks timing, one JSON line. No request for a hardware curve."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if _ndev() < 8:
        env["REEVR_BENCH_SAME_DEVICE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--cpu-seconds", "0", "--side", "0", "--watchdog", "300"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["steps"] == 2 and rec["value"] > 0
    assert rec["config"]["gather"] is True and rec["config"]["gather_matches_output"] is True
    if extra[1] == "2":
        assert rec["scaling"] == "weak"
        assert rec["config"]["channels_per_gpu"] == 4 and rec["config"]["instances_total"] == 16
        assert rec["probe"]["ok"] is True, rec["probe"]
        # whole-job aggregate: 8 ranks x 4 channels x frames per step x steps / the slowest rank's time
        frames = rec["config"]["frames_per_channel_per_step"]
        assert abs(rec["value"] - 8 * 4 * frames * 2 / (rec["ms_per_step"] * 2e-3) / 1e6) <= 1e-3 * rec["value"]
    if extra[1] == "4":
        assert rec["scaling"] == "strong"
        assert rec["config"]["channels_per_gpu"] == 2 and rec["config"]["instances_total"] == 8     # one stereo instance per rank
        assert rec["config"]["gathered_channels_per_gpu"] == 2
        assert rec["probe"]["ok"] is True, rec["probe"]
    if extra[1] == "5":
        assert rec["scaling"] == "strong"
        assert rec["config"]["channels_per_gpu"] == 8 and rec["config"]["instances_total"] == 32    # 64 mono channels = 32 pairs, 4 pairs per rank
        assert rec["config"]["gathered_channels_per_gpu"] == 8


@pytest.mark.parametrize("extra", [["--config", "4", "--blocks-per-step", "32"],
                                   ["--config", "2", "--channels", "8", "--blocks-per-step", "32"]], ids=["cfg4", "cfg2"])
def test_bench_bare_gpus_two_launches_itself(extra):
    """`python bench.py --gpus 2 ...` as the driver types it -- no launcher, no WORLD_SIZE: bench.py starts its own two ranks
    under torch.distributed.run (one device each over RCCL where two are visible; on a 1-GPU box both on device 0 over
    gloo, and the line says so) and rank 0 prints the one compact line LAST with n_gpus == 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "TORCHELASTIC_RUN_ID", "REEVR_BENCH_SAME_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--cpu-seconds", "0", "--side", "0", "--watchdog", "120"] + extra, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out_lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert out_lines[-1].startswith("{") and len(out_lines[-1]) < 4096        # the compact line is the LAST stdout line
    assert len([ln for ln in out_lines if ln.startswith("{")]) == 1
    rec = json.loads(out_lines[-1])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["config"]["gather"] is True
    assert rec["config"]["devices"] == ([0, 1] if _ndev() >= 2 else [0, 0])
    assert rec["config"]["shared_device"] == (_ndev() < 2)
    assert rec["config"]["gather_matches_output"] is True and rec["probe"]["ok"] is True


def test_bench_single_gpu_line_with_side_legs():
    """The N = 1 run as the driver types it, in small: the side legs (reference-order schedule, stereo pair, CPU baseline) run
    after the measured set has been closed -- everything the record needs from that set must have been read before -- and
    the LAST stdout line is the compact (< 4 KB) line with roofline and cpu_baseline; the full record is on disk."""
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    with tempfile.TemporaryDirectory() as d:
        full = os.path.join(d, "full.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--channels", "64",
                            "--blocks-per-step", "32", "--configs", "", "--regimes", "0", "--cpu-seconds", "2", "--full-out", full,
                            "--watchdog", "200"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out_lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert out_lines[-1].startswith("{") and len(out_lines[-1]) < 4096
        rec = json.loads(out_lines[-1])
        rec_full = json.load(open(full))
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["value"] > 0 and rec["probe"]["ok"] is True
    assert rec["roofline"]["bound"] == "hbm" and 0 < rec["roofline"]["frac"] <= 1.0 and rec["roofline"]["kernel"]
    assert rec["cpu_baseline"]["value"] > 0 and rec["cpu_baseline"]["cores"] == 1 and rec["cpu_baseline"]["kind"] in ("reference", "port")
    assert "transforms" in rec["config"] and rec["config"]["channels_per_gpu"] == 64
    assert rec["side"]["reference_schedule_Msamples_s"] > 0 and rec["side"]["stereo_pair_us_per_block"] > 0
    assert rec_full["value"] == rec["value"] and "roofline_all" in rec_full and "kernels_ms" in rec_full


def test_bench_refuses_a_mismatched_world_size():
    """Under a launcher that started a different number of ranks than --gpus says, bench.py must fail, not report a number."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE" in (r.stdout + r.stderr)


def _rccl_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import reevr_amd
    from reevr_amd import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        n_units, frames = 4, 4096
        mine = shard.units_for_rank(n_units, world, rank)
        irs = [synth.synth_ir(3000, 2, inst=u)[c] for u in mine for c in range(2)]
        x = np.stack([synth.synth_input(frames, 2 * u + c) for u in mine for c in range(2)])
        s = reevr_amd.ConvolverSet(2 * len(mine), device=rank)
        assert s.init(64, 256, irs, max_len=64)
        y = s.process_device_blocks(torch.from_numpy(x).cuda(rank), 64)
        g = shard.gather_batches(y.view(len(mine), 2, frames), dist)          # RCCL all_gather_into_tensor
        full = shard.reassemble(g, n_units, world)
        slow = shard.max_over_ranks(1.0 + rank, dist, torch.device("cuda", rank))
        q.put((rank, full.cpu().numpy(), slow))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_rccl_one_rank_bench_gather():
    """The RCCL code path of bench.py on ONE device: under torch.distributed.run with a single rank the process group is
    created with backend nccl (= RCCL) and device_id, every step's output batch goes through an asynchronous
    all_gather_into_tensor ordered behind the set's streams, the batch buffers are reused only after their gather has
    completed -- and the gathered batch of the last step must equal the set's own output; the impulse probe must hold."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for extra in (["--config", "2", "--channels", "40", "--blocks-per-step", "32", "--gather", "2"],
                  ["--config", "2", "--channels", "40", "--blocks-per-step", "32"],
                  ["--config", "4", "--blocks-per-step", "32"]):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "1",
               "--cpu-seconds", "0", "--side", "0", "--watchdog", "120"] + extra
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert rec["n_gpus"] == 1 and rec["config"]["gather"] is True
        assert rec["config"]["gather_matches_output"] is True
        assert rec["probe"]["ok"] is True, rec["probe"]


def test_rccl_gather_one_rank_against_oracle():
    """shard.gather_batches / reassemble / max_over_ranks through RCCL with a group of one rank, real engine, vs the oracle."""
    import torch.multiprocessing as mp
    from oracle import oracle_py as O
    from reevr_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(0, 1, _free_port(), q))
    p.start()
    rank, full, slow = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0 and slow == 1.0 and full.shape == (4, 2, 4096)
    for u in range(4):
        irs = synth.synth_ir(3000, 2, inst=u)
        for c in range(2):
            o = O.TwoStageFFTConvolver("orc")
            assert o.init(64, 256, irs[c])
            want = o.process(synth.synth_input(4096, 2 * u + c))
            err = np.sqrt(np.mean((full[u, c].astype(np.float64) - want) ** 2)) / np.sqrt(np.mean(want.astype(np.float64) ** 2))
            assert err <= 1e-5, (u, c, err)


def test_rccl_gather_two_devices():
    if _ndev() < 2:
        pytest.skip("needs two visible devices")
    import torch.multiprocessing as mp
    from oracle import oracle_py as O
    from reevr_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames = 4096
    for rank, full, slow in res:
        assert slow == 2.0
        assert full.shape == (4, 2, frames)
        for u in range(4):
            irs = synth.synth_ir(3000, 2, inst=u)
            for c in range(2):
                o = O.TwoStageFFTConvolver("orc")
                assert o.init(64, 256, irs[c])
                want = o.process(synth.synth_input(frames, 2 * u + c))
                err = np.sqrt(np.mean((full[u, c].astype(np.float64) - want) ** 2)) / np.sqrt(np.mean(want.astype(np.float64) ** 2))
                assert err <= 1e-5, (rank, u, c, err)
