#!/usr/bin/env python3
"""Diagnostic: the one-pass step's kernels on fans that are NOT planar in dim 2 (what a probe-pose optimisation produces;
src/renderer.py:119-124 takes any `directions`), event-timed like tools/time_step.py.  One line per fan geometry:
roll / pitch in degrees (diffus_amd.phantom.pose_ring) and the plane the ring lies in.  Library: argv[1] (default in-tree)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["DIFFUS_LIB"] = os.path.abspath(sys.argv[1])
import torch  # noqa: E402

from bench import time_events  # noqa: E402
from diffus_amd import CapturedStep, _lib  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402

N = int(os.environ.get("N", "256")); RAYS = int(os.environ.get("RAYS", "256")); SAMPLES = int(os.environ.get("SAMPLES", "512"))
P = int(os.environ.get("POSES", "32")); IT = int(os.environ.get("ITERS", "200"))
CASES = [(0, 0, (0, 1)), (5, 0, (0, 1)), (20, 0, (0, 1)), (45, 0, (0, 1)), (0, 20, (0, 1)), (20, 10, (0, 1)), (0, 0, (0, 2)),
         (0, 0, (1, 2)), (20, 0, (0, 2))]
if os.environ.get("QUICK"):      # CASES_ENV: a short list for A/B runs of library variants
    CASES = [(0, 0, (0, 1)), (5, 0, (0, 1)), (20, 0, (0, 1)), (0, 20, (0, 1))]
vol = torch.from_numpy(phantom(N)).cuda()
for roll, pitch, plane in CASES:
    src, dirs = pose_ring(N, P, RAYS, roll_deg=roll, pitch_deg=pitch, plane=plane)
    hp = CapturedStep(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), SAMPLES, 1e-4, "trilinear",
                      layout=os.environ.get("LAYOUT", "paired"))
    for _ in range(10):
        hp.step()
    scan = time_events(lambda: hp.step_mse(_lib.BWD_SCAN, epilogue=False), IT)
    scat = time_events(lambda: hp.bwd(_lib.BWD_SCATTER), IT, pre=hp.finish_grad)
    hp.finish_grad()
    flush = time_events(hp.finish_grad, IT, pre=lambda: hp.bwd(_lib.BWD_SCATTER))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        hp.step()
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 500
    print("roll %4.1f pitch %4.1f plane %s  P=%d scan %.2f us  scatter %.2f us  flush %.2f us (medians)  eager step %.2f us" % (
        roll, pitch, plane, P, scan["median"] * 1e3, scat["median"] * 1e3, flush["median"] * 1e3, step * 1e6), flush=True)
    del hp
