"""Times the scan-conversion kernels on 32 frames of 256 x 512 samples -> 256 x 256 images."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, diffus_amd
from diffus_amd.phantom import phantom, pose_ring
vol = torch.from_numpy(phantom(256)).cuda(); s, d = pose_ring(256, 32, 256)
frames, idx = diffus_amd.render_poses(vol, torch.from_numpy(s), torch.from_numpy(d), 512, 1e-4, return_indices=True)
P = 32; c0 = idx[0].reshape(P, -1).float(); c1 = idx[1].reshape(P, -1).float(); f = frames.reshape(P, -1).clone().requires_grad_(True)
g = torch.ones(P, 256, 256, device="cuda")
for _ in range(3):
    o = diffus_amd.splat_frames(c0, c1, f, 256, 256, 2.0, 512); o.backward(g)
torch.cuda.synchronize()
N = 20
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(N):
    e[0].record(); o = diffus_amd.splat_frames(c0, c1, f, 256, 256, 2.0, 512); e[1].record(); o.backward(g); e[2].record()
    torch.cuda.synchronize()
    tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
print("splat 32 frames 256x512 -> 256x256: fwd %.1f us, bwd %.1f us (mean of %d, through the Python wrapper)" % (tf / N * 1e3, tb / N * 1e3, N))
