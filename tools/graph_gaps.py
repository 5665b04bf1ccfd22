#!/usr/bin/env python3
"""Diagnostic (run under rocprofv3 --kernel-trace): replays the captured headline step; tools/graph_gaps_report.py then
prints the idle time between consecutive kernels of the replays."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffus_amd import CapturedStep  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402

P = int(os.environ.get("POSES", "32"))
vol = torch.from_numpy(phantom(256)).cuda()
src, dirs = pose_ring(256, P, 256)
hp = CapturedStep(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), 512, 1e-4, "trilinear")
g = hp.capture()
for _ in range(300):
    g.replay()
torch.cuda.synchronize()
print("ok")
