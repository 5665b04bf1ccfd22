"""world_size-2 gloo test of the N>1 path (logic only: the renderer is injected,
here the CPU oracle; on GPUs the same code runs with backend 'nccl' = RCCL)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, P, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffus_amd.distributed import render_sharded, shard_bounds
    from diffus_amd.phantom import phantom, pose_ring
    from oracle import autograd_ref as ar

    n, R, S, alpha = 32, 6, 40, 1e-3
    vol = torch.from_numpy(phantom(n)).double().requires_grad_(True)
    src, dirs = pose_ring(n, P, R)
    src_t = torch.from_numpy(src).double().requires_grad_(True)
    dirs_t = torch.from_numpy(dirs).double()

    def render_fn(v, s, d):
        return torch.stack([ar.render(v, s[p], d[p], S, alpha, 0, "trilinear") for p in range(s.shape[0])]) \
            if s.shape[0] else torch.zeros((0, R, S), dtype=torch.float64)

    frames, losses, losses_all = render_sharded(render_fn, vol, src_t, dirs_t, lambda f: (f ** 2).sum((1, 2)))
    if losses.numel():
        losses.sum().backward()
    gvol = vol.grad if vol.grad is not None else torch.zeros_like(vol)
    from diffus_amd.distributed import allreduce_volume_grad
    allreduce_volume_grad(gvol)
    lo, hi = shard_bounds(P, rank, world)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), losses_all=losses_all.numpy(), gvol=gvol.detach().numpy(),
             gsrc=(src_t.grad.numpy() if src_t.grad is not None else np.zeros((P, 3))), lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def _single(P):
    sys.path.insert(0, ROOT)
    from diffus_amd.phantom import phantom, pose_ring
    from oracle import autograd_ref as ar
    n, R, S, alpha = 32, 6, 40, 1e-3
    vol = torch.from_numpy(phantom(n)).double().requires_grad_(True)
    src, dirs = pose_ring(n, P, R)
    s = torch.from_numpy(src).double().requires_grad_(True)
    d = torch.from_numpy(dirs).double()
    losses = torch.stack([(ar.render(vol, s[p], d[p], S, alpha, 0, "trilinear") ** 2).sum() for p in range(P)])
    losses.sum().backward()
    return losses.detach().numpy(), vol.grad.numpy(), s.grad.numpy()


def _run(P, tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, P, str(tmp_path)), nprocs=world, join=True)
    ref_l, ref_gv, ref_gs = _single(P)
    outs = [np.load(os.path.join(tmp_path, f"r{r}.npz")) for r in range(world)]
    for o in outs:
        np.testing.assert_allclose(o["losses_all"], ref_l, rtol=1e-12)       # every rank sees all P losses, in order
        np.testing.assert_allclose(o["gvol"], ref_gv, rtol=1e-9, atol=1e-18)  # all-reduced shared-volume gradient
    gs = np.zeros_like(ref_gs)
    for o in outs:                                                             # pose gradients stay on the owning rank
        lo, hi = int(o["lo"]), int(o["hi"])
        gs[lo:hi] = o["gsrc"][lo:hi]
        mask = np.ones(P, bool); mask[lo:hi] = False
        assert np.all(o["gsrc"][mask] == 0)
    np.testing.assert_allclose(gs, ref_gs, rtol=1e-9)


def test_two_ranks_even_split(tmp_path):
    _run(4, tmp_path)


def test_two_ranks_ragged_split(tmp_path):
    _run(3, tmp_path)


def _sparse_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffus_amd.distributed import allreduce_box, allreduce_touched, allreduce_volume_grad
    g = torch.Generator().manual_seed(100 + rank)
    # (a) one slice of a canonical gradient
    shape = (12, 10, 8)
    gv = torch.randn(shape, generator=g)
    dense = gv.clone()
    box = ((0, 12), (0, 10), (3, 4))
    moved_box = allreduce_box(gv, box)
    allreduce_volume_grad(dense)
    # (b) the bricked scratch + touched flags of a scatter: each rank touches its own random third of 60 bricks
    nb = 60
    touched = (torch.rand(nb, generator=g) < 0.33).to(torch.int32)
    touched[7] = 2 if rank == 0 else 0                          # a flag 2 ("stale", left by a PERSISTENT flush) is not live scratch
    bricks = torch.randn(nb, 32, generator=g) * (touched == 1).unsqueeze(1)
    want = bricks.clone()
    allreduce_volume_grad(want)
    t_before = touched.clone()
    moved_t = allreduce_touched(bricks.view(-1), touched)
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), gv=gv.numpy(), dense=dense.numpy(), own=torch.randn(shape, generator=torch.Generator().manual_seed(100 + rank)).numpy(),
             bricks=bricks.numpy(), want=want.numpy(), touched=touched.numpy(), t_before=t_before.numpy(), moved_box=moved_box, moved_t=moved_t)
    dist.barrier()
    dist.destroy_process_group()


def test_sparse_allreduce_of_a_shared_volume_gradient(tmp_path):
    """allreduce_box / allreduce_touched against the dense all_reduce of the whole tensor (world 2, gloo): same sums where a
    step can have written, nothing else touched, a fraction of the bytes (SURVEY §8e "Collective"; the reference's training
    loop learns one slice: `[DEMO] Train MRI to Impedance MLP - GPU` cell 16)."""
    world, port = 2, _free_port()
    mp.spawn(_sparse_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(os.path.join(tmp_path, f"s{r}.npz")) for r in range(world)]
    union = (outs[0]["t_before"] != 0) | (outs[1]["t_before"] != 0)
    for r, o in enumerate(outs):
        np.testing.assert_allclose(o["gv"][:, :, 3], o["dense"][:, :, 3], rtol=1e-6)        # the slice: summed over the ranks
        rest = np.ones(o["gv"].shape, bool); rest[:, :, 3] = False
        assert np.array_equal(o["gv"][rest], o["own"][rest])                                 # everything else: this rank's own values
        assert int(o["moved_box"]) == 12 * 10 * 4
        np.testing.assert_allclose(o["bricks"], o["want"], rtol=1e-6, atol=1e-7)             # bricked scratch: the dense sum
        assert np.array_equal(o["touched"] != 0, union)                                     # every rank flushes the union
        assert np.all(o["touched"][(o["t_before"] == 0) & union] == 1)
        assert int(o["moved_t"]) == 60 + int(union.sum()) * 32 * 4 < 60 * 32 * 4
