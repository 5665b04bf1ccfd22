#!/usr/bin/env python3
"""Forward launch of BASELINE config 3 (32 poses x 256 rays x 512 steps, 256^3, trilinear, paired), event-timed, and how many of its
rays carry a large |echo| (the float64 re-evaluation threshold, DESIGN fact 45).  python tools/time_fwd.py [library.so]"""
import sys, os
if len(sys.argv) > 1:
    os.environ["DIFFUS_LIB"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import diffus_amd as da
from diffus_amd.phantom import phantom, pose_ring
vol = torch.from_numpy(phantom(256)).cuda()
s, d = pose_ring(256, 32, 256)
s = torch.from_numpy(s).cuda(); d = torch.from_numpy(d).cuda()
bv = da.pair_volume(vol) if hasattr(da, "pair_volume") else None
with torch.no_grad():
    for _ in range(20):
        f = da.render_poses(vol, s, d, 512, 1e-4, sampler="trilinear", layout="paired")
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        f = da.render_poses(vol, s, d, 512, 1e-4, sampler="trilinear", layout="paired")
    e1.record(); torch.cuda.synchronize()
echo = f * torch.exp(1e-4 * torch.arange(512, device="cuda"))
em = echo.abs().amax(dim=2)
print(os.path.basename(os.environ.get("DIFFUS_LIB", "in-tree")), "forward 32 poses: %.2f us per launch; rays with |echo| > 8: %d, > 4: %d, > 3: %d, > 2: %d, > 1.5: %d, > 1: %d of %d" % (
    e0.elapsed_time(e1) / 300 * 1e3, int((em > 8).sum()), int((em > 4).sum()), int((em > 3).sum()), int((em > 2).sum()), int((em > 1.5).sum()), int((em > 1).sum()), em.numel()))
