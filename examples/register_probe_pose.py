#!/usr/bin/env python3
"""Six-degree-of-freedom probe registration on one full-size frame (BASELINE config 2: 256 rays x 512 steps, 256^3).

What `notebooks/[NW] alignement.ipynb` cells 13-14 of the reference set out to do -- move `source` / `directions` until the
rendered frame matches an observed one -- and cannot: `plot_beam_frame` (src/renderer.py:201-275) rounds the sample points
(:754-756), so no gradient reaches the pose.  Here the trilinear sampler carries d loss / d source and
d loss / d directions out of the HIP backward, and `FanPose` carries them on to apex, median angle and rotation vector.

    python examples/register_probe_pose.py [iterations] [--graph] [--one-pass] [--poses P]

--graph: the whole iteration (FanPose -> render -> loss -> backward -> Adam) captured once as a HIP graph and replayed; every
launch of it is capturable (nothing allocates behind torch's back or synchronises), and the loop is launch-bound otherwise.
--one-pass: render + loss + backward as `CapturedStep.mse_loss` -- frame, loss and the pose gradients out of ONE pass over the
samples (diffus_render_step_mse) instead of a forward launch, three loss kernels and a backward that recomputes the forward.
--poses P: a SWEEP of P frames registered together (probe positions on a ring around the head, each 3 voxels and 5 degrees off):
one FanPose module with (P,3) apexes, one render launch per iteration for all of them.

The observed frame is rendered from a "true" pose; the start pose is 3 voxels and 5 degrees (roll + pitch, out of the slice)
away.  Prints the loss, the apex error and the worst ray angle as the descent goes, and the time per iteration.
"""
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffus_amd as da  # noqa: E402


def smooth_head(n):
    """A head-like volume WITHOUT hard edges: registration by gradient descent needs a loss that varies smoothly with the
    pose, and the phantom's skull (400 -> 6.4e6 in one voxel) makes the frame a step function of it."""
    u = np.arange(n, dtype=np.float64) / (n - 1)
    c = u - 0.5
    r2 = (c / 0.46)[:, None, None] ** 2 + (c / 0.40)[None, :, None] ** 2 + (c / 0.44)[None, None, :] ** 2
    vol = 1.5e6 + 2.5e5 * np.exp(-3.0 * r2) + 1.2e5 * (np.sin(19 * u)[:, None, None] * np.cos(17 * u)[None, :, None]
                                                       * np.sin(13 * u + 1)[None, None, :])
    return vol.astype(np.float32)


def worst_ray_angle(pose, true):
    with torch.no_grad():
        a, b = pose()[1].double().cpu().reshape(-1, 3), true()[1].double().cpu().reshape(-1, 3)
    cosang = (a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))
    return float(torch.rad2deg(torch.acos(cosang.clamp(-1, 1))).max())


def apex_error(pose, true):
    """Largest apex distance over the poses, voxels."""
    return float(torch.linalg.norm((pose.apex.detach() - true.apex.detach()).reshape(-1, 3), dim=1).max())


def run(iters=400, n=256, R=256, S=512, alpha=1e-4, report=50, graph=False, quiet=False, stats=None, one_pass=False, poses=1):
    say = (lambda *a: None) if quiet else print
    vol = torch.from_numpy(smooth_head(n)).cuda()
    P = int(poses)
    phi = np.arctan2(0.6, 0.8) + 2.0 * np.pi * np.arange(P) / P                   # pose 0 looks along (0.8, 0.6)
    look = np.stack([np.cos(phi), np.sin(phi), np.zeros(P)], 1)
    side = np.stack([-np.sin(phi), np.cos(phi), np.zeros(P)], 1)
    apex_true = np.array([0.5 * n, 0.5 * n, 0.5 * n])[None, :] - 0.30 * n * look
    tilt = np.radians(4.0) * look + np.radians(3.0) * side                        # 5 degrees in all, out of the slice both ways
    off = np.array([1.8, -1.9, 1.5])[None, :] * np.where(np.arange(P)[:, None] % 2 == 0, 1.0, -1.0)   # 3.0 voxels, alternating sides
    one = (lambda a: a[0]) if P == 1 else (lambda a: a)                            # one pose: (3,) / (2,) parameters, (R,3) directions
    true = da.FanPose(one(apex_true), one(look[:, :2]), math.radians(60.0), R, rotvec=one(np.zeros((P, 3)))).cuda()
    with torch.no_grad():
        target = da.render_poses(vol, *true(), S, alpha, sampler="trilinear")
    pose = da.FanPose(one(apex_true + off), one(look[:, :2]), math.radians(60.0), R, rotvec=one(tilt)).cuda()
    opt = torch.optim.Adam([{"params": [pose.apex], "lr": 0.05}, {"params": [pose.median_angle, pose.rotvec], "lr": 0.002}],
                           fused=True, capturable=graph)
    say("start: apex error %.2f voxels, worst ray angle %.2f deg" % (apex_error(pose, true), worst_ray_angle(pose, true)))
    loss_out = torch.zeros((), device="cuda")

    step = None
    if one_pass:   # persistent buffers, no volume gradient (the volume is not what is being learnt), the launch for any fan
        with torch.no_grad():
            s0, d0 = pose()
        step = da.CapturedStep(vol, s0.detach().reshape(P, 3).clone(), d0.detach().reshape(P, R, 3).clone(), S, alpha, "trilinear",
                               want_gvol=False, fans="oblique", target=target, loss_scale=1.0)

    def iteration():
        opt.zero_grad(set_to_none=True)
        src, dirs = pose()
        if step is not None:
            loss = step.mse_loss(sources=src, directions=dirs)
            loss.backward(step.unit)       # (the step's resident 1.0: the gradients are handed over as they are)
        else:
            frame = da.render_poses(vol, src, dirs, S, alpha, sampler="trilinear")
            loss = ((frame - target) ** 2).sum()
            loss.backward()
        opt.step()
        loss_out.copy_(loss.detach())

    done = 0
    replay = iteration
    if graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # torch's recipe: a few eager iterations on a side stream first
            for _ in range(3):
                iteration()
        torch.cuda.current_stream().wait_stream(side)
        done = 3
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            iteration()
        done += 1
        replay = g.replay
    history = []
    torch.cuda.synchronize()
    t0, paused = time.perf_counter(), 0.0
    for it in range(done, iters):
        replay()
        if it % report == 0 or it == iters - 1:
            torch.cuda.synchronize()
            tp = time.perf_counter()                       # (the clock stops for a report: it renders, copies to the host and prints)
            history.append((it, float(loss_out)))
            say("iter %4d  loss %.4g  apex error %.3f voxels  worst ray angle %.3f deg" % (
                it, history[-1][1], apex_error(pose, true), worst_ray_angle(pose, true)))
            torch.cuda.synchronize()
            paused += time.perf_counter() - tp
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0 - paused
    say("%d iterations of %d pose(s), %.3f ms each (FanPose + render fwd + bwd + Adam%s; reports not counted)" % (
        iters - done, P, 1e3 * dt / max(iters - done, 1), ", one graph replay" if graph else ""))
    if stats is not None:
        stats["ms_per_iteration"] = 1e3 * dt / max(iters - done, 1)
    return history, apex_error(pose, true), worst_ray_angle(pose, true)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("iterations", nargs="?", type=int, default=400)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--one-pass", action="store_true")
    ap.add_argument("--poses", type=int, default=1)
    a = ap.parse_args()
    run(a.iterations, graph=a.graph, one_pass=a.one_pass, poses=a.poses)
