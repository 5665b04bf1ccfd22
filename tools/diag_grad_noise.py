#!/usr/bin/env python3
"""How close to its tolerance does tests/test_hip_parity.py::test_backward_vs_float64_autograd sit?  Prints the frame and
gradient errors of its cases for the library given as argv[1] (a diagnostic build under variants/, or the product)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    os.environ["DIFFUS_LIB"] = os.path.abspath(sys.argv[1])
import diffus_amd as da  # noqa: E402
from conftest import maxnorm_rel  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402
from oracle import autograd_ref as ar  # noqa: E402

n = 64
vol_np = phantom(n)
for sampler in ("nearest", "trilinear"):
    for S, start in ((48, 0), (150, 0), (300, 12), (513, 0), (1024, 0)):
        src, dirs = pose_ring(n, 4, 6)
        src, dirs = src[1], dirs[1].copy()
        dirs[:, 2] = 0.21
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        if S > 300:
            dirs *= np.float32(40.0 / S)
        alpha = 3e-3
        v = torch.from_numpy(vol_np).double().requires_grad_(True)
        s = torch.from_numpy(src).double().requires_grad_(sampler == "trilinear")
        d = torch.from_numpy(dirs).double().requires_grad_(sampler == "trilinear")
        f = ar.render(v, s, d, S, alpha, start, sampler, points="f32")
        g = torch.Generator().manual_seed(1)
        up = torch.randn(f.shape, generator=g, dtype=torch.float64)
        (f * up).sum().backward()
        out = []
        for layout in ("canonical", "paired"):
            vol = torch.from_numpy(vol_np).cuda().requires_grad_(True)
            sc = torch.from_numpy(src).cuda().requires_grad_(True)
            dc = torch.from_numpy(dirs).cuda().requires_grad_(True)
            fr = da.render_poses(vol, sc, dc, S, alpha, start=start, sampler=sampler, layout=layout)[0]
            (fr * up.float().cuda()).sum().backward()
            e = [maxnorm_rel(fr.detach().cpu().numpy(), f.detach().numpy()), maxnorm_rel(vol.grad.cpu().numpy(), v.grad.numpy())]
            if sampler == "trilinear":
                e += [maxnorm_rel(sc.grad.cpu().numpy(), s.grad.numpy()), maxnorm_rel(dc.grad.cpu().numpy(), d.grad.numpy())]
            out.append(layout + " " + " ".join("%.2e" % x for x in e))
        print(f"{sampler:9s} S={S:4d} start={start:2d}  frame gvol [gsrc gdirs]:  " + "   ".join(out))
