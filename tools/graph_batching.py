#!/usr/bin/env python3
"""How the captured headline step should be issued: eager launches, one hipGraph per step, or several steps per graph
(the idle time between two graph launches is ~8.6 us on this stack; between kernels of one graph it is 0)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffus_amd import CapturedStep  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402

P = int(os.environ.get("POSES", "32"))
vol = torch.from_numpy(phantom(256)).cuda()
src, dirs = pose_ring(256, P, 256)
hp = CapturedStep(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), 512, 1e-4, "trilinear")
for _ in range(5):
    hp.step()
torch.cuda.synchronize()


def wall(fn, steps_per_call, total=480):
    n = total // steps_per_call
    for _ in range(max(2, 16 // steps_per_call)):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * steps_per_call) * 1e6


print("eager launches          %.2f us per step" % wall(hp.step, 1))
side = torch.cuda.Stream()
for m in (1, 2, 4, 8, 16):
    g = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=side):
        for _ in range(m):
            hp.step()
    print("graph of %2d step(s)      %.2f us per step" % (m, wall(g.replay, m)))
