"""Multi-GPU layer: poses shard across ranks, one collective for the losses.

The reference has no distributed code at all; this is the north
star's "batches of probe poses shard embarrassingly across the 8 GPUs of one
node with a single RCCL gather".  One process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the logic tests).

  * poses are split contiguously: rank g renders poses [lo_g, hi_g);
  * a pose's rays are never split (the start>0 median couples them, and 256 rays
    are too few to split anyway);
  * the volume is replicated (or distinct per rank, BASELINE config 5);
  * forward data path: no collective.  Afterwards ONE all_gather of the P
    per-pose losses (P floats).  Only when the shared volume itself is being
    learned is there a second collective: one all_reduce(SUM) of its gradient.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(P_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first P_total % world ranks get one extra pose."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(P_total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_losses(local: torch.Tensor, P_total: int, group=None) -> torch.Tensor:
    """All ranks get the (P_total,) vector of per-pose losses, in pose order.
    Equal shards use one all_gather_into_tensor; ragged shards pad to the max."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(P_total, g, world) for g in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    assert local.numel() == counts[rank], (local.numel(), counts[rank])
    local = local.contiguous()
    if len(set(counts)) == 1:
        out = torch.empty(P_total, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    m = max(counts)
    pad = torch.zeros(m, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    buf = torch.empty(world * m, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[g * m: g * m + counts[g]] for g in range(world)])


def allreduce_volume_grad(gvol: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank gradients of a SHARED (replicated) volume, in place."""
    dist.all_reduce(gvol, op=dist.ReduceOp.SUM, group=group)
    return gvol


def render_sharded(render_fn: Callable, volume, sources: torch.Tensor, directions: torch.Tensor,
                   loss_fn: Callable[[torch.Tensor], torch.Tensor], group=None,
                   reduce_volume_grad: Optional[torch.Tensor] = None):
    """Render this rank's shard of the P poses and gather the per-pose losses.

    render_fn(volume, sources_shard, directions_shard) -> frames (p,R,N1)   e.g. a partial of
        diffus_amd.render_poses; tests inject the CPU oracle here.
    loss_fn(frames) -> (p,) per-pose losses.
    Returns (frames_local, losses_local, losses_all).  The backward of
    losses_local.sum() gives this rank's pose gradients (they stay local) and its
    contribution to the volume gradient; pass that tensor as `reduce_volume_grad`
    on a later call to all-reduce it.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    P_total = sources.shape[0]
    lo, hi = shard_bounds(P_total, rank, world)
    d = directions[lo:hi] if directions.dim() == 3 else directions
    frames = render_fn(volume, sources[lo:hi], d)
    losses = loss_fn(frames)
    losses_all = gather_losses(losses.detach(), P_total, group)
    if reduce_volume_grad is not None:
        allreduce_volume_grad(reduce_volume_grad, group)
    return frames, losses, losses_all
