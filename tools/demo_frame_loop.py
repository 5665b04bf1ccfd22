#!/usr/bin/env python3
"""The REUBEN demo frame through the drop-in API, 300 times (for rocprofv3 --kernel-trace: which kernels a frame is, how
long they take, how far apart they start).  Prints the wall time per frame."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch, diffus_amd
from diffus_amd.phantom import phantom
vol = torch.from_numpy(phantom(256)).cuda()
source = torch.tensor([88.0769, -11.5385, 110.0], dtype=torch.float64)
dirs = diffus_amd.generate_cone_directions(np.array([0.35, 0.94]), np.radians(52.47), 256)
rend = diffus_amd.UltrasoundRenderer(num_samples=185, attenuation_coeff=1e-4)
def frame():
    x, y, z, I = rend.plot_beam_frame(volume=vol, source=source, directions=dirs, plot=False, artifacts=True, start=40, seed=0)
    return diffus_amd.differentiable_splat(x, y, z, I, H=256, W=256, sigma=1)
for _ in range(10): frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for _ in range(N): frame()
torch.cuda.synchronize()
print("wall per frame: %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
