"""Times one training step through the Python mirror (torch autograd): render_poses -> sum of squares -> backward, for
the volume and the pose, against the direct C-ABI step bench.py measures."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, diffus_amd
from diffus_amd.phantom import phantom, pose_ring
vol = torch.from_numpy(phantom(256)).cuda().requires_grad_(True)
for P in (32, 1):
    s, d = pose_ring(256, 32, 256)
    s = torch.from_numpy(s[:P]).cuda().requires_grad_(True); d = torch.from_numpy(d[:P]).cuda().requires_grad_(True)
    def step():
        f = diffus_amd.render_poses(vol, s, d, 512, 1e-4, sampler="trilinear")
        (f * f).sum().backward()
        vol.grad = None; s.grad = None; d.grad = None
    for _ in range(5): step()
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for _ in range(N): step()
    torch.cuda.synchronize()
    print("P=%d: %.1f us per step through autograd (wall)" % (P, (time.perf_counter() - t0) / N * 1e6))
