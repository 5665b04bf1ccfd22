#!/usr/bin/env python3
"""Diagnostic: frames of the forward kernel and of the one-pass (adjoint-scan) kernel against float64, for a library build
(DIFFUS_LIB), on the small-step long-ray case of tests/test_handback.py.  usage: tools/diag_lerp_order.py LIB"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DIFFUS_LIB"] = os.path.abspath(sys.argv[1])
import numpy as np, torch
from diffus_amd import CapturedStep
from diffus_amd.phantom import phantom, pose_ring
from oracle import autograd_ref as ar
n, P, R, alpha = 64, 3, 20, 1e-3
vol_np = phantom(n)
vol = torch.from_numpy(vol_np).cuda()
src, dirs = pose_ring(n, 8, R)
for S, start, sc in ((700, 30, 0.08), (513, 0, 1.0), (300, 12, 1.0), (1027, 3, 0.08)):
    s = torch.from_numpy(src[:P]).cuda()
    dn = (dirs[:P] * sc).astype(np.float32)
    d = torch.from_numpy(dn).cuda().contiguous()
    two = CapturedStep(vol, s, d, S, alpha, "trilinear", start=start, persistent=False, one_pass=False); two.step()
    one = CapturedStep(vol, s, d, S, alpha, "trilinear", start=start, persistent=False); one.step()
    torch.cuda.synchronize()
    f64 = np.stack([ar.render(torch.from_numpy(vol_np).double(), torch.from_numpy(src[p]).double(), torch.from_numpy(dn[p]).double(),
                              S, alpha, start, "trilinear", points="f32").numpy() for p in range(P)])
    den = np.abs(f64).max()
    a, b = two.frame.cpu().numpy(), one.frame.cpu().numpy()
    print("S=%d start=%d step=%.2f  max|f|=%.1f  fwd-kernel vs f64 %.2e   one-pass vs f64 %.2e   fwd vs one-pass %.2e" %
          (S, start, sc, den, np.abs(a - f64).max() / den, np.abs(b - f64).max() / den, np.abs(a - b).max() / den))
