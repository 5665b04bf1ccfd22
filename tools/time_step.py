#!/usr/bin/env python3
"""Diagnostic: the one-pass step's kernels for a library build (argv[1]), event-timed over many launches (median / min),
and the eager step as a whole.  Workload from POSES / N / RAYS / SAMPLES / LAYOUT (default: BASELINE config 3)."""
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
P = int(os.environ.get("POSES", "32")); IT = int(os.environ.get("ITERS", "300"))
vol_np = phantom(N)
if os.environ.get("ZERO_BG"):       # a masked volume -- zero background, Z_{n-1} + Z_n == 0 along every ray outside the head
    vol_np[vol_np == 400.0] = 0.0
vol = torch.from_numpy(vol_np).cuda()
src, dirs = pose_ring(N, P, RAYS, roll_deg=float(os.environ.get("ROLL", "0")), pitch_deg=float(os.environ.get("PITCH", "0")))
hp = CapturedStep(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), SAMPLES, 1e-4, "trilinear",
                  layout=os.environ.get("LAYOUT", "paired"))
for _ in range(20):
    hp.step()
scan = time_events(lambda: hp.step_mse(_lib.BWD_SCAN, epilogue=False), IT)
scat = time_events(lambda: hp.bwd(_lib.BWD_SCATTER), IT, pre=hp.finish_grad)
hp.finish_grad()
flush = time_events(hp.finish_grad, IT, pre=lambda: hp.bwd(_lib.BWD_SCATTER))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(1000):
    hp.step()
torch.cuda.synchronize()
step = (time.perf_counter() - t0)
print("%-28s P=%d scan %.2f / %.2f us  scatter %.2f / %.2f us  flush %.2f / %.2f us (median / min)  eager step %.2f us" % (
    os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "in-tree", P, scan["median"] * 1e3, scan["min"] * 1e3, scat["median"] * 1e3,
    scat["min"] * 1e3, flush["median"] * 1e3, flush["min"] * 1e3, step * 1e3))
