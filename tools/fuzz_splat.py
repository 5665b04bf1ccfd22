#!/usr/bin/env python3
"""Fuzz differentiable_splat (diffus_splat_fwd / _bwd; reference src/renderer.py:694-737) against the NumPy oracle (oracle/splat.py,
itself pinned by golden G11 = the reference run): random sample grids (integer index planes like plot_beam_frame's, rotated float
coordinates, points far outside the image, many samples per pixel), image sizes, sigmas, intensities; forward <= 2e-6, gradient <= 2e-5
(max-norm relative: the bars of tests/test_splat.py).

    python tools/fuzz_splat.py [first_seed] [count]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffus_amd as da  # noqa: E402
from oracle import splat as osp  # noqa: E402


def rel(a, b):
    m = float(np.max(np.abs(b)))
    return float(np.max(np.abs(a - b))) / (m if m > 0 else 1.0)


def gen(seed):
    rng = np.random.default_rng(40000 + seed)
    R, N = int(rng.integers(1, 70)), int(rng.integers(1, 200))
    H, W = int(rng.integers(4, 300)), int(rng.integers(4, 300))
    sigma = float(rng.choice([0.1, 0.3, 0.5, 1.0, 1.0, 2.0, 2.0, 3.7, 6.0, 8.0]))
    kind = seed % 4
    # a fan in some plane: two varying coordinates, one (nearly) constant -- the axis choice goes by variance
    ang = rng.uniform(0, 2 * np.pi) + np.linspace(-0.5, 0.5, R)[:, None]
    rad = np.arange(N)[None, :] * rng.uniform(0.2, 2.0)
    u = rng.uniform(0, W) + rad * np.cos(ang)
    v = rng.uniform(0, H) + rad * np.sin(ang)
    if kind == 3:      # far outside: everything clamps onto the border
        u += 5 * W
    c = np.full((R, N), float(rng.integers(0, 50)))
    planes = [u, v, c]
    perm = rng.permutation(3)
    x, y, z = (planes[i] for i in perm)
    if kind in (0, 3):                       # integer index planes (what plot_beam_frame returns)
        x, y, z = (np.rint(a).astype(np.int64) for a in (x, y, z))
    else:                                    # rotated float coordinates (rotate_around_apex's output)
        x, y, z = (a.astype(np.float32) for a in (x, y, z))
    f = rng.standard_normal((R, N)).astype(np.float32) * (10.0 if kind == 2 else 1.0)
    up = rng.standard_normal((W, H)).astype(np.float32)
    return x, y, z, f, up, H, W, sigma


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    bad, wf, wg = 0, 0.0, 0.0
    t0 = time.time()
    for k, seed in enumerate(range(first, first + count)):
        x, y, z, f, up, H, W, sigma = gen(seed)
        o, _ = osp.splat(x, y, z, f, H, W, sigma)
        go = osp.splat_grad(x, y, z, f, up, H, W, sigma)
        ft = torch.from_numpy(f).cuda().requires_grad_(True)
        why = None
        try:
            out = da.differentiable_splat(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(z).cuda(), ft, H=H, W=W, sigma=sigma)
            (out * torch.from_numpy(up).cuda()).sum().backward()
            ef, eg = rel(out.detach().cpu().numpy(), o), rel(ft.grad.cpu().numpy(), go)
            wf, wg = max(wf, ef), max(wg, eg)
            if out.shape != (W, H):
                why = "shape %s" % (tuple(out.shape),)
            elif not (ef < 2e-6 and eg < 2e-5):
                why = "forward %.2e gradient %.2e" % (ef, eg)
        except Exception as e:      # noqa: BLE001
            why = "raised %r" % (e,)
        if why:
            bad += 1
            print("FAIL seed %d grid %dx%d image %dx%d sigma %.1f %s: %s" % (seed, x.shape[0], x.shape[1], H, W, sigma, x.dtype, why), flush=True)
        if (k + 1) % 250 == 0:
            print("%d cases, %d failures, worst forward %.2e gradient %.2e, %.0f s" % (k + 1, bad, wf, wg, time.time() - t0), flush=True)
    print("done: %d cases from %d, %d failures" % (count, first, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
