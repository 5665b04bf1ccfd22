#!/usr/bin/env python3
"""Diagnostic: where the HOST time of one drop-in plot_beam_frame call goes (cProfile, 2000 calls, device drained every 50)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, diffus_amd
from diffus_amd.phantom import phantom
vol = torch.from_numpy(phantom(256)).cuda()
source = torch.tensor([88.0769, -11.5385, 110.0], dtype=torch.float64)
dirs = diffus_amd.generate_cone_directions(np.array([0.35, 0.94]), np.radians(52.47), 256)
rend = diffus_amd.UltrasoundRenderer(num_samples=185, attenuation_coeff=1e-4)
def frame():
    return rend.plot_beam_frame(volume=vol, source=source, directions=dirs, plot=False, artifacts=False, start=40)
for _ in range(50): frame()
torch.cuda.synchronize()
N = 2000
t0 = time.perf_counter()
for i in range(N):
    frame()
    if i % 50 == 49: torch.cuda.synchronize()
torch.cuda.synchronize()
print("wall per call %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
pr = cProfile.Profile(); pr.enable()
for i in range(N):
    frame()
    if i % 50 == 49: torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
