#!/usr/bin/env python3
"""Where a 6-DoF registration iteration (examples/register_probe_pose.py) spends its host time: cProfile over the loop, and the
loop's parts timed one by one with the device drained in between.  python tools/prof_registration.py"""
import cProfile
import math
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import diffus_amd as da  # noqa: E402
from register_probe_pose import smooth_head  # noqa: E402

n, R, S, alpha = 256, 256, 512, 1e-4
vol = torch.from_numpy(smooth_head(n)).cuda()
look = np.array([0.8, 0.6, 0.0])
apex = np.array([0.5 * n] * 3) - 0.30 * n * look
true = da.FanPose(apex, look[:2], math.radians(60.0), R, rotvec=(0.0, 0.0, 0.0)).cuda()
with torch.no_grad():
    target = da.render_poses(vol, *true(), S, alpha, sampler="trilinear")
pose = da.FanPose(apex + np.array([1.8, -1.9, 1.5]), look[:2], math.radians(60.0), R, rotvec=(0.05, 0.04, 0.0)).cuda()
opt = torch.optim.Adam([{"params": [pose.apex], "lr": 0.05}, {"params": [pose.median_angle, pose.rotvec], "lr": 0.002}], fused=True)


def step():
    opt.zero_grad(set_to_none=True)
    src, dirs = pose()
    frame = da.render_poses(vol, src, dirs, S, alpha, sampler="trilinear")
    loss = ((frame - target) ** 2).sum()
    loss.backward()
    opt.step()


for _ in range(30):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print("iteration: %.3f ms" % (1e3 * (time.perf_counter() - t0) / 200))


def timed(label, fn, reps=200):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print("  %-44s %.3f ms" % (label, 1e3 * (time.perf_counter() - t) / reps))
    return out


src, dirs = timed("FanPose forward", lambda: pose())
frame = timed("render_poses forward (grad mode)", lambda: da.render_poses(vol, src, dirs, S, alpha, sampler="trilinear"))
loss = timed("loss", lambda: ((frame - target) ** 2).sum())
timed("backward (loss -> pose parameters)", lambda: loss.backward(retain_graph=True))
timed("Adam step", lambda: opt.step())
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
