#!/usr/bin/env python3
"""Diagnostic driver for rocprofv3: a few EAGER one-pass steps of the headline workload (or POSES/N/RAYS/SAMPLES from the
environment) with the library given as argv[1] (default: the in-tree build).  Nothing is timed here."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["DIFFUS_LIB"] = os.path.abspath(sys.argv[1])
import torch  # noqa: E402

from diffus_amd import CapturedStep  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402

N = int(os.environ.get("N", "256")); RAYS = int(os.environ.get("RAYS", "256")); SAMPLES = int(os.environ.get("SAMPLES", "512"))
P = int(os.environ.get("POSES", "32")); STEPS = int(os.environ.get("STEPS", "6"))
vol = torch.from_numpy(phantom(N)).cuda()
src, dirs = pose_ring(N, P, RAYS, roll_deg=float(os.environ.get("ROLL", "0")), pitch_deg=float(os.environ.get("PITCH", "0")))
hp = CapturedStep(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), SAMPLES, 1e-4,
                  os.environ.get("SAMPLER", "trilinear"), layout=os.environ.get("LAYOUT", "paired"))
for _ in range(STEPS):
    hp.step()
torch.cuda.synchronize()
print("ok", float(hp.loss.sum()))
