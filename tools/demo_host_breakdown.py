#!/usr/bin/env python3
"""Host time of the pieces of the drop-in demo frame (perf_counter around each call, device not waited for)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch, diffus_amd
from diffus_amd.phantom import phantom
from diffus_amd import splat as sp
vol = torch.from_numpy(phantom(256)).cuda()
source = torch.tensor([88.0769, -11.5385, 110.0], dtype=torch.float64)
dirs = diffus_amd.generate_cone_directions(np.array([0.35, 0.94]), np.radians(52.47), 256)
rend = diffus_amd.UltrasoundRenderer(num_samples=185, attenuation_coeff=1e-4)
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
def frame():
    t = time.perf_counter()
    x, y, z, I = rend.plot_beam_frame(volume=vol, source=source, directions=dirs, plot=False, artifacts=False, start=40, seed=0)
    tick("plot_beam_frame(artifacts=False)", t); t = time.perf_counter()
    I2 = diffus_amd.apply_artifacts(I, seed=0)
    tick("apply_artifacts", t); t = time.perf_counter()
    sel, _ = sp.select_axes(x, y, z, vol.device)
    tick("select_axes", t); t = time.perf_counter()
    out = sp.splat_frames(sel[0:1], sel[1:2], I2.reshape(1, -1), 256, 256, 1.0, I2.shape[-1])
    tick("splat_frames", t)
    return out
for _ in range(20): frame()
torch.cuda.synchronize(); T.clear()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    frame()
    if _ % 20 == 19: torch.cuda.synchronize()       # keep the queue short: host times without back-pressure
torch.cuda.synchronize()
print("wall per frame %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
for k, v in T.items():
    print("  %-36s %6.1f us" % (k, v / N * 1e6))
print("  %-36s %6.1f us" % ("sum", sum(T.values()) / N * 1e6))
