#!/usr/bin/env python3
"""Diagnostic: does splitting the pose batch into two independent scan->scatter chains on two streams hide the kernels'
ramp and tail?  Times (wall, 1000 iterations) the scan+scatter pair of one POSES-pose step against two POSES/2-pose
steps issued on two streams."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffus_amd import CapturedStep, _lib  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402

N = int(os.environ.get("N", "256")); RAYS = int(os.environ.get("RAYS", "256")); SAMPLES = int(os.environ.get("SAMPLES", "512"))
P = int(os.environ.get("POSES", "32")); IT = int(os.environ.get("ITERS", "1000"))
vol = torch.from_numpy(phantom(N)).cuda()
src, dirs = pose_ring(N, P, RAYS)
src, dirs = torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda()


def mk(lo, hi):
    return CapturedStep(vol, src[lo:hi].contiguous(), dirs[lo:hi].contiguous(), SAMPLES, 1e-4, "trilinear", layout="paired")


def wall(fn):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(IT):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / IT * 1e6


whole = mk(0, P)
t_whole = wall(lambda: whole.step_mse(_lib.BWD_ALL))
halves = [mk(0, P // 2), mk(P // 2, P)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def two():
    for h, s in zip(halves, streams):
        with torch.cuda.stream(s):
            h.step_mse(_lib.BWD_ALL)


t_two = wall(two)
# host-only cost of the same calls (one stream, same work serialised)
t_two_serial = wall(lambda: [h.step_mse(_lib.BWD_ALL) for h in halves])
# the whole step: both chains scatter into ONE gradient scratch, flushed once after a join
halves[1].gvol_k, halves[1].touched, halves[1].gvol = halves[0].gvol_k, halves[0].touched, halves[0].gvol
main = torch.cuda.current_stream()
side = streams[1]
e_flush, e_side = torch.cuda.Event(), torch.cuda.Event()


def step_two():
    side.wait_event(e_flush)
    with torch.cuda.stream(side):
        halves[1].step_mse(_lib.BWD_ALL)
        e_side.record(side)
    halves[0].step_mse(_lib.BWD_ALL)
    main.wait_event(e_side)
    halves[0].finish_grad()
    e_flush.record(main)


e_flush.record(main)
t_step_two = wall(step_two)
t_step = wall(whole.step)
print("P=%d  whole step: one chain %.2f us   two chains %.2f us" % (P, t_step, t_step_two))
print("P=%d  scan+scatter: one chain %.2f us   two chains on two streams %.2f us   two chains on one stream %.2f us" % (
    P, t_whole, t_two, t_two_serial))
