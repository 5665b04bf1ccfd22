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
    learned is there a second collective: one all_reduce(SUM) of its gradient --
    of the PART of it a step can have touched (allreduce_box: the slice the
    reference's training loop learns, 256 KiB at 256^3; allreduce_touched: the
    union of the bricks the ranks' scatters added into), not of all 64 MiB.
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


def allreduce_box(gvol: torch.Tensor, box, group=None, scratch: Optional[torch.Tensor] = None) -> int:
    """Sum, in place, the part of a shared volume's canonical gradient inside `box` = ((x0, x1), (y0, y1), (z0, z1)), half-open --
    e.g. `CapturedStep.dirty_box`'s slice: the reference's training loop (`[DEMO] Train MRI to Impedance MLP - GPU` cell 16)
    learns ONE dim-2 slice of the volume, so only that slice of d/dvolume is ever read.  The box is packed into `scratch` (or a
    new tensor), all-reduced (one collective) and written back; returns the bytes that went through the collective (256 KiB
    for a slice of 256^3 against the dense call's 64 MiB: 0.3-0.75 ms over xGMI, several 0.056 ms steps)."""
    (x0, x1), (y0, y1), (z0, z1) = ((int(a), int(b)) for a, b in box)
    view = gvol[x0:x1, y0:y1, z0:z1]
    n = view.numel()
    if n == 0:
        return 0
    buf = scratch[:n].view(view.shape) if scratch is not None else torch.empty(view.shape, dtype=gvol.dtype, device=gvol.device)
    buf.copy_(view)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    view.copy_(buf)
    return n * gvol.element_size()


def allreduce_touched(gvol_bricked: torch.Tensor, touched: torch.Tensor, group=None) -> int:
    """Sum, in place, a shared volume's BRICKED gradient scratch over the ranks, moving only bricks some rank's scatter added
    into: the touched-brick flags (one per 4 x 4 x 2 brick, set by the scatter kernel) are OR-ed over the ranks (all_reduce(MAX)
    of one byte per brick: 512 KiB at 256^3), the union's bricks are packed, summed (one all_reduce) and written back, and every
    rank's flags become the union -- so that each rank's diffus_gradbuf_flush hands back the same summed gradient.  Call it
    between the scatter and the flush (CapturedStep.bwd(BWD_SCATTER) / finish_grad()).  Returns the bytes through the two
    collectives.  A 32-pose step touches ~80 000 of the 524 288 bricks of a 256^3 volume: 10 MB instead of 64 MiB.
    (One host round trip for the size of the union: `nonzero`.)"""
    nb = touched.numel()
    bricks = gvol_bricked.view(nb, -1)
    flags = (touched != 0).to(torch.uint8)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)        # (NCCL has no bitwise OR: a byte per brick, MAX)
    idx = flags.nonzero(as_tuple=True)[0]
    moved = flags.numel() * flags.element_size()
    if idx.numel():
        packed = bricks.index_select(0, idx)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        bricks.index_copy_(0, idx, packed)
        # a brick this rank did not touch itself: flag 1 (live scratch: the flush must convert it), as the scatter would have set
        touched.masked_fill_((flags != 0) & (touched == 0), 1)
        moved += packed.numel() * packed.element_size()
    return moved


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
