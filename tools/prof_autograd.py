"""Host-time profile of one training step through the drop-in autograd path (render_poses -> loss -> backward)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cProfile, pstats, time, torch, diffus_amd
from diffus_amd.phantom import phantom, pose_ring
vol = torch.from_numpy(phantom(256)).cuda().requires_grad_(True)
s, d = pose_ring(256, 32, 64)
s = torch.from_numpy(s[:1]).cuda().requires_grad_(True); d = torch.from_numpy(d[:1]).cuda().requires_grad_(True)
def step():
    f = diffus_amd.render_poses(vol, s, d, 228, 1e-4, start=110, sampler="trilinear")
    (f * f).sum().backward()
    vol.grad = None; s.grad = None; d.grad = None
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize()
print("wall per step: %.1f us" % ((time.perf_counter() - t0) / 100 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
