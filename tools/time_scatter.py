#!/usr/bin/env python3
"""Diagnostic: event-time the three hot kernels for a given library build (DIFFUS_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DIFFUS_LIB"] = os.path.abspath(sys.argv[1])
import torch
from diffus_amd import CapturedStep as HotPath
from bench import time_events
from diffus_amd import _lib
from diffus_amd.phantom import phantom, pose_ring
N = int(os.environ.get("N", "256")); RAYS = int(os.environ.get("RAYS", "256")); SAMPLES = int(os.environ.get("SAMPLES", "512"))
vol = torch.from_numpy(phantom(N)).cuda()
P = int(os.environ.get("POSES", "32"))
src, dirs = pose_ring(N, P, RAYS)
hp = HotPath(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), SAMPLES, 1e-4, "trilinear")
hp.fwd(); hp.loss_and_grad(); hp.zero_grad(); hp.bwd(_lib.BWD_SCAN)
for _ in range(3):
    hp.step()
f = time_events(hp.fwd, 30)["median"]
b = time_events(lambda: hp.bwd(_lib.BWD_SCAN), 30)["median"]
s = time_events(lambda: hp.bwd(_lib.BWD_SCATTER), 30, pre=hp.finish_grad)["median"]
u = time_events(hp.finish_grad, 30, pre=lambda: hp.bwd(_lib.BWD_SCATTER))["median"]
o = time_events(lambda: hp.step_mse(_lib.BWD_SCAN), 30)["median"]
g = hp.capture()
t = time_events(g.replay, 50)["median"]
print("P=%d " % P + "%-34s fwd %.1f us  bwd-scan %.1f us  scatter %.1f us  flush %.1f us  one-pass scan %.1f us  step (graph) %.1f us" % (os.path.basename(sys.argv[1]), f * 1e3, b * 1e3, s * 1e3, u * 1e3, o * 1e3, t * 1e3))
