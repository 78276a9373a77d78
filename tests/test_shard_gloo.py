"""World-size-2 test of the multi-GPU path on CPU (gloo): static unit->rank map, independent
processing per rank with no data-path collective, max-over-ranks timing, and the optional
gather of output batches. The compute stand-in is the CPU oracle (the checker), so what is
tested is exactly the sharding/gather logic bench.py uses on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_units, frames, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle_py as O
    from reevr_amd import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard.units_for_rank(n_units, world, rank)
        outs = []
        for u in mine:                       # one stereo instance per unit, its own IR and input
            irs = synth.synth_ir(700, 2, inst=u)
            y = []
            for c in range(2):
                conv = O.TwoStageFFTConvolver("orc")
                assert conv.init(32, 128, irs[c])
                y.append(conv.process(synth.synth_input(frames, c + 2 * u)))
            outs.append(np.stack(y))
        local = torch.from_numpy(np.stack(outs))
        g = shard.gather_batches(local, dist)
        full = shard.reassemble(g, n_units, world)
        slow = shard.max_over_ranks(1.0 + rank, dist)
        q.put((rank, mine, full.numpy(), slow))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_units_map():
    from reevr_amd import shard
    assert shard.units_for_rank(8, 8, 3) == [3]
    assert shard.units_for_rank(8, 2, 1) == [1, 3, 5, 7]
    assert shard.units_for_rank(64, 8, 0) == list(range(0, 64, 8))
    seen = sorted(u for r in range(4) for u in shard.units_for_rank(8, 4, r))
    assert seen == list(range(8))
    assert all(shard.owner_of(u, 4) == r for r in range(4) for u in shard.units_for_rank(8, 4, r))
    with pytest.raises(ValueError):
        shard.units_for_rank(8, 2, 2)


def test_world2_gloo_shard_and_gather():
    import torch.multiprocessing as mp
    from oracle import oracle_py as O
    from reevr_amd import synth
    O.backend("orc")                         # build the oracle once, before forking workers
    world, n_units, frames = 2, 4, 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process answer
    want = []
    for u in range(n_units):
        irs = synth.synth_ir(700, 2, inst=u)
        y = []
        for c in range(2):
            conv = O.TwoStageFFTConvolver("orc")
            assert conv.init(32, 128, irs[c])
            y.append(conv.process(synth.synth_input(frames, c + 2 * u)))
        want.append(np.stack(y))
    want = np.stack(want)
    for rank, mine, full, slow in res:
        assert mine == list(range(rank, n_units, world))
        assert np.array_equal(full, want)    # every rank holds every unit's output, in unit order
        assert slow == 2.0                   # max over ranks of (1 + rank)
