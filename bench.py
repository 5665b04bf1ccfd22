#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DiffUS hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): ray-steps/s, forward + backward, 256^3 volume, 256 rays x
512 steps.  One *step* = one pass of the hot path over one batch of poses:
  forward frame (diffus_render_fwd)
  -> loss_p = sum(frame_p^2), dL/dframe = 2 frame (diffus_loss_sumsq)
  -> backward (diffus_render_bwd: d/d volume, d/d source, d/d directions)
  -> touched bricks of the gradient scratch -> the caller's canonical (d0,d1,d2) volume-gradient tensor
     (diffus_gradbuf_flush, mode PERSISTENT: the tensor is kept across steps and bricks the previous step
     wrote but this one does not are cleared, so after every step it IS that step's dense gradient -- the
     64 MiB memset per step this replaces is still available as --memset-grad)
  -> gather of the per-pose losses over ranks (RCCL, N > 1).
Workload at N=1 = BASELINE config 3 (32 poses of the config-2 shape on one GPU;
a single 256x512 frame is only 256 wavefronts, i.e. launch-latency-bound, and is
reported separately as `single_pose`).  Multi-GPU: poses shard contiguously over
ranks, 32 per GPU (weak scaling; N=8 is BASELINE config 4), the volume is
replicated, the only collective is one all_gather of P losses.

Inputs (volume -- canonical and its bricked copy --, poses) are resident in HBM
before the timed region.  The step is issued either eagerly or, by default, as a
captured hipGraph replay (the C-ABI never syncs or allocates, so it captures).
Kernel durations for the roofline come from HIP events recorded on the launch
stream (torch's current stream, which is the one handed to the C-ABI).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from diffus_amd import _lib  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402

# Algorithmic bytes per ray-step, no-reuse model (SURVEY §8d / DESIGN.md §Roofline), per kernel:
#   fwd      8 corner reads x 4 B + 4 B frame write                         = 36 (nearest: 4 + 4 = 8)
#   bwd scan 4 B gframe read + 8 x 4 B corner re-reads                      = 36 (nearest: 8)
#   scatter  8 corners x (4 B read + 4 B write) atomic RMW on the gradient  = 64 (nearest: 8)
BYTES = {
    "trilinear": {"render_fwd_kernel": 36, "render_bwd_kernel": 36, "scatter_patch_kernel": 64},
    "nearest": {"render_fwd_kernel": 8, "render_bwd_kernel": 8, "scatter_patch_kernel": 8},
}
HBM_PEAK_GBS = 8000.0
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class HotPath:
    """Pre-allocated buffers + direct C-ABI calls (what a captured training step does)."""

    def __init__(self, vol, src, dirs, S, alpha, sampler, start=0, want_gvol=True, layout="paired", sparse=True,
                 persistent=True):
        self.lib = _lib.load()
        self.vol, self.src, self.dirs = vol, src, dirs
        self.layout = {"canonical": 0, "bricked": 1, "paired": 2}[layout]
        self.P, self.R = dirs.shape[0], dirs.shape[1]
        self.S, self.start, self.alpha = S, start, alpha
        self.sampler = {"nearest": 0, "trilinear": 1}[sampler]
        dev = vol.device
        self.N1 = S - start
        self.frame = torch.empty((self.P, self.R, self.N1), dtype=torch.float32, device=dev)
        self.gframe = torch.empty_like(self.frame)
        d0, d1, d2 = vol.shape
        self.dims = (d0, d1, d2)
        # canonical gradient, what the caller gets.  persistent: the tensor is kept across steps and
        # diffus_gradbuf_flush(PERSISTENT) clears what the previous step left where this step adds nothing, so it
        # always equals this step's dense gradient without a 64 MiB memset per step.
        self.persistent = persistent and sparse and want_gvol and self.layout != 0
        self.gvol = torch.zeros_like(vol) if want_gvol else None
        if self.layout != 0:
            # HBM-resident converted copy of the (constant) volume, made once outside the timed region;
            # the bricked gradient scratch is zeroed, filled and converted back EVERY step.
            nb = self.lib.diffus_bricked_floats(d0, d1, d2)
            if self.layout == 1:
                self.vol_k = torch.empty(nb, dtype=torch.float32, device=dev)
                _lib.check(self.lib.diffus_brick_volume(vp(vol), d0, d1, d2, vp(self.vol_k), self.stream()), "brick")
            else:
                self.vol_k = torch.empty(self.lib.diffus_paired_floats(d0, d1, d2), dtype=torch.float32, device=dev)
                _lib.check(self.lib.diffus_pair_volume(vp(vol), d0, d1, d2, vp(self.vol_k), self.stream()), "pair")
            # sparse gradient hand-back: the bricked scratch and its touched-brick flags are all-zero
            # between steps (diffus_gradbuf_flush restores that), only touched bricks are converted
            self.gvol_k = torch.zeros(nb, dtype=torch.float32, device=dev) if want_gvol else None
            self.touched = (torch.zeros(self.lib.diffus_brick_count(d0, d1, d2), dtype=torch.int32, device=dev)
                            if (want_gvol and sparse) else None)
        else:
            self.vol_k, self.gvol_k, self.touched = vol, self.gvol, None
        self.gsrc = torch.empty((self.P, 3), dtype=torch.float32, device=dev)
        self.gdirs = torch.empty((self.P, self.R, 3), dtype=torch.float32, device=dev)
        self.loss = torch.empty((self.P,), dtype=torch.float32, device=dev)
        self.loss_ws = torch.zeros(max(512 * self.P, 512), dtype=torch.uint8, device=dev)   # arrival counters: zero once
        nws = max(self.lib.diffus_workspace_bytes(self.P, self.R, S, start), 256)
        self.ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        self.common = (vp(self.vol_k), d0, d1, d2, self.layout, vp(src), 0, vp(dirs), 0, self.P, self.R, S, start,
                       float(alpha), self.sampler)

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fwd(self):
        _lib.check(self.lib.diffus_render_fwd(*self.common, vp(self.frame), None, vp(self.ws), self.ws.numel(),
                                              self.stream()), "diffus_render_fwd")

    def bwd(self, stages=_lib.BWD_ALL):
        _lib.check(self.lib.diffus_render_bwd(*self.common, vp(self.gframe), vp(self.gvol_k), vp(self.touched),
                                              vp(self.gsrc), vp(self.gdirs), stages, vp(self.ws), self.ws.numel(),
                                              self.stream()), "diffus_render_bwd")

    def loss_and_grad(self):
        _lib.check(self.lib.diffus_loss_sumsq(vp(self.frame), self.P, self.R * self.N1, vp(self.loss),
                                              vp(self.gframe), vp(self.loss_ws), self.loss_ws.numel(), self.stream()),
                   "diffus_loss_sumsq")

    def zero_grad(self):
        """A fresh dense gradient every step: zero the caller's canonical (d0,d1,d2) tensor (sparse
        hand-back), or the bricked scratch (dense hand-back: the conversion overwrites every voxel)."""
        if self.gvol is not None and not self.persistent:
            (self.gvol if (self.touched is not None or self.layout == 0) else self.gvol_k).zero_()

    def finish_grad(self):
        """touched bricks of the scratch -> added into the canonical gradient; scratch back to all-zero."""
        if self.layout != 0 and self.gvol is not None and self.touched is not None:
            # accumulate = 0: the tensor was zeroed this step and every touched voxel is written once
            _lib.check(self.lib.diffus_gradbuf_flush(vp(self.gvol_k), vp(self.touched), *self.dims, vp(self.gvol),
                                                     2 if self.persistent else 0, self.stream()), "diffus_gradbuf_flush")
        elif self.layout != 0 and self.gvol is not None:
            _lib.check(self.lib.diffus_unbrick_volume(vp(self.gvol_k), *self.dims, vp(self.gvol), 0, self.stream()),
                       "diffus_unbrick_volume")

    def step(self):
        # (a forked stream for zero_grad beside the forward was measured: the fork/join events cost
        # more than the 10 us they hide -- 0.217 vs 0.202 ms/step -- so the step stays on one stream)
        self.fwd()
        self.loss_and_grad()
        self.zero_grad()
        self.bwd()
        self.finish_grad()


def time_events(fn, iters, pre=None):
    """Device time of fn() in ms (mean, median, min): HIP events on the current (launch) stream."""
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        if pre is not None:
            pre()
        e0[i].record()
        fn()
        e1[i].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in zip(e0, e1))
    return {"mean": sum(ts) / len(ts), "median": ts[len(ts) // 2], "min": ts[0]}


def cpu_baseline(budget_rays=64, budget_steps=256):
    """The reference's algorithm (N+1 dense torch.linalg.solve, oracle/dense.py) on
    the host cores, on a bounded sample: config 1 = 64 rays x 256 steps, forward
    only (the dense backward needs ~rays*steps^3 memory)."""
    from oracle import dense
    from oracle import oracle as orc
    n = 256
    vol = phantom(n)
    src, dirs = pose_ring(n, 32, budget_rays)
    # the GPU box gives one GPU a 16-core share of the host; more threads than that (or than the
    # affinity mask) only oversubscribes the many small LAPACK calls
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    dense.plot_beam_frame_dense(torch.from_numpy(vol), torch.from_numpy(src[0]), torch.from_numpy(dirs[0]),
                                budget_steps, 1e-4, 0)
    dt = time.perf_counter() - t0
    out = {"value": budget_rays * budget_steps / dt, "unit": "ray-steps/s", "cores": torch.get_num_threads(),
           "kind": "port",
           "sample": f"reference algorithm (N+1 dense torch.linalg.solve, oracle/dense.py), forward only, "
                     f"1 pose x {budget_rays} rays x {budget_steps} steps on the 256^3 phantom "
                     f"(BASELINE config 1), {dt:.2f} s wall"}
    # algorithm-matched extra: the O(N) scalar C oracle, one 256x512 pose, trilinear forward
    src2, dirs2 = pose_ring(n, 32, 256)
    orc.build()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 2.0:
        orc.plot_beam_frame(vol, src2[reps % 32], dirs2[reps % 32], 512, 1e-4, 0, sampler="trilinear")
        reps += 1
    dt2 = time.perf_counter() - t0
    out["cpu_scan_value"] = reps * 256 * 512 / dt2
    out["cpu_scan_note"] = "O(N) running-product C oracle (oracle/diffus_oracle.c), 1 thread, trilinear forward only"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--poses", type=int, default=32, help="poses per GPU")
    ap.add_argument("--rays", type=int, default=256)
    ap.add_argument("--samples", type=int, default=512)
    ap.add_argument("--n", type=int, default=256, help="volume edge")
    ap.add_argument("--sampler", default="trilinear", choices=["trilinear", "nearest"])
    ap.add_argument("--no-gvol", action="store_true", help="pose-gradient-only backward")
    ap.add_argument("--layout", default="paired", choices=["paired", "bricked", "canonical"])
    ap.add_argument("--memset-grad", action="store_true",
                    help="zero the whole canonical gradient tensor every step instead of keeping it persistent "
                         "(diffus_gradbuf_flush mode STORE instead of PERSISTENT)")
    ap.add_argument("--dense-grad", action="store_true",
                    help="hand the gradient back by a dense conversion instead of the touched-brick flush")
    ap.add_argument("--eager", action="store_true", help="issue launches from Python instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--alpha", type=float, default=1e-4)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 logic on a single GPU (every rank on cuda:0)")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N > 1: issue the loss all_gather on the compute stream (default: on its own stream, overlapping "
                         "the next step's kernels)")
    ap.add_argument("--force-dist", action="store_true",
                    help="debug: initialise torch.distributed even for a single rank, to exercise the N > 1 code path")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    ngpu = world
    dev = torch.device("cuda", torch.cuda.current_device())

    P_total = args.poses * ngpu
    # config 5 (--n 512): one distinct volume per GPU -- the phantom variant (tumour position) is the rank
    vol = torch.from_numpy(phantom(args.n, variant=rank if args.n >= 512 else 0)).to(dev)
    src_all, dirs_all = pose_ring(args.n, P_total, args.rays)
    lo = rank * args.poses
    src = torch.from_numpy(src_all[lo:lo + args.poses]).to(dev).contiguous()
    dirs = torch.from_numpy(dirs_all[lo:lo + args.poses]).to(dev).contiguous()
    hp = HotPath(vol, src, dirs, args.samples, args.alpha, args.sampler, want_gvol=not args.no_gvol,
                 layout=args.layout, sparse=not args.dense_grad, persistent=not args.memset_grad)
    losses_all = torch.empty((P_total,), dtype=torch.float32, device=dev)

    # --- the step: eager launches, or one captured hipGraph (compute) + the collective ---
    # Two loss buffers, used alternately (and one captured graph per buffer): with N > 1 the gather of step k reads
    # its buffer on the communication stream while step k+1 already writes the other one.
    loss_buf = [hp.loss, torch.empty_like(hp.loss)]
    graph = None
    graphs = [None, None]
    side = torch.cuda.Stream()
    if not args.eager:
        try:
            with torch.cuda.stream(side):
                for _ in range(3):
                    hp.step()
            side.synchronize()
            for b in range(2):
                hp.loss = loss_buf[b]
                graphs[b] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graphs[b], stream=side):
                    hp.step()
            graph = graphs[0]
        except Exception as e:  # capture unsupported -> eager, and say so
            print(f"hipGraph capture failed ({e!r}); running eagerly", file=sys.stderr)
            graph = None
        hp.loss = loss_buf[0]

    # The one collective of the path: all_gather of the P per-pose losses (P x 4 bytes per rank) over xGMI.  A small
    # RCCL collective is tens of microseconds of pure latency -- a quarter of the 0.14 ms step -- so by default it
    # runs on its own stream, reading step k's loss buffer while step k+1 writes the other one (an event per buffer
    # orders reuse).  Every gather has completed before the closing barrier of the timed region.
    overlap = dist is not None and args.dist_backend == "nccl" and not args.sync_gather
    if overlap:
        try:
            comm = torch.cuda.Stream()
            gathered = [torch.empty((P_total,), dtype=torch.float32, device=dev) for _ in range(2)]
            step_ev = [torch.cuda.Event() for _ in range(2)]
            gather_ev = [torch.cuda.Event() for _ in range(2)]
        except Exception as e:
            print(f"overlapped gather unavailable ({e!r}); gathering on the compute stream", file=sys.stderr)
            overlap = False
    kstep = [0]

    def step():
        b = kstep[0] & 1 if overlap else 0
        if overlap:
            main = torch.cuda.current_stream()
            if kstep[0] >= 2:
                main.wait_event(gather_ev[b])          # loss buffer b is free again (gather k-2 has finished)
            hp.loss = loss_buf[b]
        if graph is not None:
            graphs[b].replay()
        else:
            hp.step()
        if overlap:
            kstep[0] += 1
            step_ev[b].record(main)
            with torch.cuda.stream(comm):
                comm.wait_event(step_ev[b])
                dist.all_gather_into_tensor(gathered[b], loss_buf[b])
                gather_ev[b].record(comm)
        elif dist is not None and args.dist_backend == "nccl":
            dist.all_gather_into_tensor(losses_all, hp.loss)
        elif dist is not None:                                  # gloo rehearsal: through host memory
            out = torch.empty(P_total, dtype=torch.float32)
            dist.all_gather_into_tensor(out, hp.loss.cpu())
            losses_all.copy_(out)

    def barrier():
        if overlap:
            torch.cuda.current_stream().wait_stream(comm)       # every gather issued so far is part of the step count
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_ms = (time.perf_counter() - t0) * 1e3 / args.steps   # host time to ENQUEUE one step (no device wait)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if overlap:
            losses_all.copy_(gathered[(kstep[0] - 1) & 1])
        # every rank must hold all P losses, in pose order, and they must be finite
        assert torch.isfinite(losses_all).all() and float(losses_all.abs().min()) > 0, "loss gather failed"
    ray_steps = P_total * args.rays * args.samples
    value = ray_steps * args.steps / dt

    # --- per-kernel device time (HIP events on the launch stream), rank-local ---
    it = max(10, min(args.steps, 50))
    hp.fwd(); hp.loss_and_grad()
    k_ms = {"render_fwd_kernel": time_events(hp.fwd, it),
            "render_bwd_kernel": time_events(lambda: hp.bwd(_lib.BWD_SCAN), it)}
    if not args.no_gvol:
        k_ms["scatter_patch_kernel"] = time_events(lambda: hp.bwd(_lib.BWD_SCATTER), it,
                                                   pre=(hp.finish_grad if hp.touched is not None else hp.zero_grad))
        hp.finish_grad()
    local_rs = args.poses * args.rays * args.samples
    b = BYTES[args.sampler]
    dom = max(k_ms, key=lambda k: k_ms[k]["mean"])
    dom_ms = k_ms[dom]["mean"]
    achieved = b[dom] * local_rs / (dom_ms * 1e-3) / 1e9
    traffic = None
    if os.path.exists(PMC_SUMMARY):      # HBM bytes per launch from the committed PMC passes (profiles/)
        try:
            pm = json.load(open(PMC_SUMMARY))
            if pm.get("workload_ray_steps") == local_rs and pm.get("sampler") == args.sampler:
                traffic = pm["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    # --- single-pose latency (BASELINE config 2): 1 pose, fwd + bwd, eager and graph-replayed ---
    hp1 = HotPath(vol, src[:1].contiguous(), dirs[:1].contiguous(), args.samples, args.alpha, args.sampler,
                  want_gvol=not args.no_gvol, layout=args.layout, sparse=not args.dense_grad,
                  persistent=not args.memset_grad)
    for _ in range(5):
        hp1.step()
    sp = time_events(hp1.step, 20)
    sp_graph = None
    try:
        with torch.cuda.stream(side):
            hp1.step()
        side.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=side):
            hp1.step()
        sp_graph = time_events(g1.replay, 20)
    except Exception:
        pass
    # the same frame with the pose-gradient-only backward that config 2 names (no d/dvolume: no scatter, no flush)
    hp1p = HotPath(vol, src[:1].contiguous(), dirs[:1].contiguous(), args.samples, args.alpha, args.sampler,
                   want_gvol=False, layout=args.layout)
    for _ in range(5):
        hp1p.step()
    sp_pose = time_events(hp1p.step, 20)

    if rank == 0:
        grads = "d/dsource, d/ddirections" if args.no_gvol else "d/dvolume, d/dsource, d/ddirections"
        out = {
            "metric": "ray-steps/sec fwd+bwd",
            "value": value,
            "unit": "ray-steps/s",
            "n_gpus": ngpu,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"BASELINE config {'3' if ngpu == 1 else '4-style'}: {args.poses} poses/GPU x "
                             f"{args.rays} rays x {args.samples} steps through a {args.n}^3 analytic head phantom; "
                             f"{args.sampler} sampling; forward + sum-of-squares loss + backward ({grads}) "
                             f"+ canonical gradient + per-pose loss gather"),
                "poses_per_gpu": args.poses, "poses_total": P_total, "rays": args.rays, "samples": args.samples,
                "volume": [args.n] * 3, "sampler": args.sampler, "start": 0, "alpha": args.alpha,
                "layout": args.layout, "grad_handback": "dense" if args.dense_grad else ("sparse (touched bricks), persistent tensor" if hp.persistent else "sparse (touched bricks), memset per step"), "issue": "hipGraph replay" if graph is not None else "eager",
                "parallelism": f"poses sharded x{ngpu}, volume replicated", "host_enqueue_ms_per_step": host_ms,
                "loss_gather": ("none (1 GPU)" if dist is None else
                                ("all_gather on its own stream, overlapping the next step" if overlap else "all_gather on the compute stream")),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "bytes_per_ray_step": b[dom],
                "ray_steps_per_launch": local_rs,
                "launch_ms": dom_ms,
                "kernels_ms": k_ms,
                "whole_step_algorithmic_GBs": (sum(b[k] for k in k_ms) * local_rs) / (dt / args.steps) / 1e9,
            },
            "single_pose": {"workload": "BASELINE config 2: 1 pose x 256 rays x 512 steps, fwd+bwd",
                            "eager_ms": sp["median"], "graph_ms": sp_graph["median"] if sp_graph else None,
                            "pose_gradient_only_ms": sp_pose["median"],
                            "value": args.rays * args.samples / (min(sp["median"], (sp_graph or sp)["median"]) * 1e-3)},
        }
        if ngpu == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is informative; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "ray-steps/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
