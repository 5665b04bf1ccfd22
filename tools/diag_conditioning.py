#!/usr/bin/env python3
"""Per-pose frame error of the HIP kernels at the benchmark workload against the float64 echo series, next to the float32
oracle's own error: which rays are ill-conditioned (grazing the skull: echo = b/d with d -> 0) and how far each float32
evaluation -- forward kernel, one-pass step, scalar C oracle -- sits from float64 there."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffus_amd as da  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402
from oracle import oracle as orc  # noqa: E402

N, P, R, S, A = 256, 32, 256, 512, 1e-4
vol_np = phantom(N)
src, dirs = pose_ring(N, P, R)
vol = torch.from_numpy(vol_np).cuda()
s, d = torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda()
f_fwd = da.render_poses(vol, s, d, S, A, sampler="trilinear", layout="paired").cpu().numpy()
step = da.CapturedStep(vol, s, d, S, A, "trilinear", layout="paired")
step.step()
torch.cuda.synchronize()
f_one = step.frame.cpu().numpy()
att = np.exp(-A * np.arange(S))


def mr(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())


print("pose  fwd-vs-64  onepass-vs-64  orc32-vs-64  fwd-vs-orc32  max|frame|")
for p in range(P):
    imp = orc.sample_trilinear(vol_np, src[p], dirs[p], S)
    r = orc.reflection(imp)
    f64 = orc.echo_scan(r.astype(np.float64), np.float64) * att
    f32 = orc.plot_beam_frame(vol_np, src[p], dirs[p], S, A, 0, sampler="trilinear")[3]
    print(f"{p:3d}  {mr(f_fwd[p], f64):.2e}  {mr(f_one[p], f64):.2e}  {mr(f32, f64):.2e}  {mr(f_fwd[p], f32.astype(np.float64)):.2e}  {np.abs(f64).max():.3g}")
