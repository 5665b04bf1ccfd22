"""Times the GPU artifact chain on 32 frames of 256 x 512 samples, and the NumPy/SciPy restatement of
the reference's chain (oracle/artifacts.py) on one frame on the host beside it."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, diffus_amd
from diffus_amd.phantom import phantom, pose_ring
vol = torch.from_numpy(phantom(256)).cuda(); s, d = pose_ring(256, 32, 256)
frames = diffus_amd.render_poses(vol, torch.from_numpy(s), torch.from_numpy(d), 512, 1e-4)
for _ in range(3):
    out = diffus_amd.apply_artifacts(frames, seed=1)
torch.cuda.synchronize()
N = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(N):
    out = diffus_amd.apply_artifacts(frames, seed=i)
e1.record(); torch.cuda.synchronize()
print("artifact chain, 32 frames 256x512 (f64): %.1f us per batch = %.2f us per frame" % (e0.elapsed_time(e1) / N * 1e3, e0.elapsed_time(e1) / N * 1e3 / 32))
from oracle import artifacts as oa
f = frames[0].cpu().numpy()
rng = np.random.default_rng(0)
rs, ls = oa.noise_scales(f.shape[1], 0.01, 0.15)
radial = rng.normal(1.0, rs); local = rng.normal(1.0, ls[None, :], size=f.shape)
t0 = time.perf_counter(); oa.chain(f, 0.01, 0.15, 4.0, 5.0, radial, local) if hasattr(oa, "chain") else None; t = time.perf_counter() - t0
if hasattr(oa, "chain"):
    print("host NumPy restatement, one frame: %.1f ms" % (t * 1e3))
