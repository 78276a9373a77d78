"""Multi-GPU sharding of independent convolver units.

The path has no cross-unit arithmetic: every `Convolver` (one IR channel x one input channel)
is independent (reference src/dsp/StereoConvolver.cpp:33-42), so units are dealt to ranks
statically, `rank = unit mod world`, at instance granularity (the 2-4 channels of one
stereo/quad instance stay on one GPU). No collective is on the data path; the optional
gather of output batches is one all_gather per *batch* of blocks, never per 512-frame block.
One process per GPU; on ROCm torch.distributed's "nccl" backend is RCCL (xGMI), "gloo" on CPU.
"""
from __future__ import annotations

from typing import List


def units_for_rank(n_units: int, world: int, rank: int) -> List[int]:
    """Static map rank = unit mod world."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_units, world))


def owner_of(unit: int, world: int) -> int:
    return unit % world


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """Slowest rank's time (bench contract: timing is the MAX over ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_batches(local, dist=None):
    """all_gather of equally-shaped per-rank output batches -> (world, *local.shape).
    `local` is a torch tensor (CUDA with nccl/RCCL, CPU with gloo)."""
    import torch
    if dist is None or not dist.is_initialized():
        return local.unsqueeze(0)
    world = dist.get_world_size()          # (a group of one rank still runs the collective)
    local = local.contiguous()
    if local.is_cuda and dist.get_backend() == "nccl":      # RCCL: one collective straight into the stacked tensor
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out.view(world * local.shape[0], *local.shape[1:]), local)
        return out
    parts = [torch.empty_like(local) for _ in range(world)]  # gloo (CPU tensors, or CUDA tensors staged by gloo)
    dist.all_gather(parts, local)
    return torch.stack(parts)


def gather_batches_async(local, out, dist=None):
    """Non-blocking form: all_gather of `local` (contiguous, equally shaped on every rank) into the preallocated
    `out` (world, *local.shape); returns a work handle (`.wait()`), or None when there is nothing to communicate.
    With RCCL the collective is ordered behind the caller's current CUDA stream and runs on the communicator's own
    stream, so the next batch's kernels (on another stream) overlap it."""
    if dist is None or not dist.is_initialized():
        out[0].copy_(local)
        return None
    world = dist.get_world_size()                         # (a group of ONE rank still runs the collective: the RCCL path
                                                          #  -- communicator, stream hand-off, buffer reuse -- on a 1-GPU box)
    assert out.shape[0] == world and tuple(out.shape[1:]) == tuple(local.shape) and local.is_contiguous()
    if local.is_cuda and dist.get_backend() == "nccl":
        return dist.all_gather_into_tensor(out.view(world * local.shape[0], *local.shape[1:]), local, async_op=True)
    return dist.all_gather([out[r] for r in range(world)], local, async_op=True)


def reassemble(gathered, n_units: int, world: int):
    """gathered[r][j] holds unit units_for_rank(n_units, world, r)[j]; return them in unit order.
    Requires n_units % world == 0 (equal shards)."""
    import torch
    assert n_units % world == 0
    per = n_units // world
    out = [None] * n_units
    for r in range(world):
        for j, u in enumerate(units_for_rank(n_units, world, r)):
            assert j < per
            out[u] = gathered[r][j]
    return torch.stack(out)
