#!/usr/bin/env python3
"""Fuzz the two auxiliary kernels of the training loop against plain torch: ImpedanceEstimator (diffus_mlp_fwd / _bwd; reference
src/impedance.py:16-17) against the same layers run by torch, and ssim_loss (diffus_ssim_loss_fwd / _bwd; the loss of
`[DEMO] Train MRI to Impedance MLP - GPU` cell 16) against examples/losses.py's torch ops -- the bars of tests/test_impedance.py and
tests/test_losses.py over random shapes, weights, masks of zeros and ties.

    python tools/fuzz_aux.py [first_seed] [count]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import diffus_amd as da  # noqa: E402
from losses import minmax01, ssim  # noqa: E402


def rel(a, b):
    m = float(b.abs().max())
    return float((a - b).abs().max()) / (m if m > 0 else 1.0)


def mlp_case(seed):
    g = torch.Generator().manual_seed(70000 + seed)
    torch.manual_seed(70000 + seed)
    m = da.ImpedanceEstimator(1).cuda()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(float(torch.empty(1).uniform_(0.3, 3.0, generator=g)))
    shape = [(int(torch.randint(1, 6000, (1,), generator=g)),), (int(torch.randint(1, 40, (1,), generator=g)), int(torch.randint(1, 90, (1,), generator=g))),
             (3, int(torch.randint(1, 50, (1,), generator=g)), 17)][seed % 3]
    x = (torch.randn(*shape, generator=g) * float(torch.empty(1).uniform_(0.1, 5.0, generator=g))).cuda()
    xr = x.clone().requires_grad_(True)
    y = m(xr)
    up = torch.randn(y.shape, generator=g).cuda()
    (y * up).sum().backward()
    xt = x.clone().requires_grad_(True)
    yt = m.model(xt.reshape(-1, 1)).reshape(x.shape)
    gt = torch.autograd.grad((yt * up).sum(), [xt] + list(m.parameters()))
    errs = {"y": rel(y.detach(), yt.detach()), "gx": rel(xr.grad, gt[0])}
    for i, (p, gr) in enumerate(zip(m.parameters(), gt[1:])):
        errs["gp%d" % i] = rel(p.grad, gr)
    # (the parameter gradients are sums over all voxels: float32 accumulation order, n up to 6000)
    ok = errs["y"] < 5e-6 and errs["gx"] < 1e-5 and all(v < 5e-5 for k, v in errs.items() if k.startswith("gp"))
    return ok, "mlp shape %s: %s" % (tuple(shape), {k: "%.1e" % v for k, v in errs.items()})


def ssim_case(seed):
    g = torch.Generator().manual_seed(80000 + seed)
    H, W = int(torch.randint(11, 330, (1,), generator=g)), int(torch.randint(11, 330, (1,), generator=g))
    normalise = seed % 2 == 0
    ref = torch.rand(H, W, generator=g).cuda()
    img = torch.rand(H, W, generator=g) * float(torch.empty(1).uniform_(0.5, 4.0, generator=g)) - (0.5 if seed % 3 == 0 else 0.0)
    if seed % 4 == 1:
        img[: H // 3] = 0.0
    if normalise:
        img = img.clamp_min(0.0)
    if seed % 5 == 2:
        img[H // 2, W // 2] = 5.0
    a = img.cuda().requires_grad_(True)
    b = img.cuda().requires_grad_(True)
    la = da.ssim_loss(a, ref, normalise=normalise)
    xb = minmax01(b) if normalise else b
    lb = 1.0 - ssim(xb[None, None], ref[None, None], data_range=1.0)
    la.backward()
    lb.backward()
    dv = abs(float(la) - float(lb))
    den = float(b.grad.abs().max())
    dg = float((a.grad - b.grad).abs().max()) / (den if den > 0 else 1.0)
    return (dv <= 3e-6 and dg <= 2e-4), "ssim %dx%d normalise %s: value %.1e gradient %.1e" % (H, W, normalise, dv, dg)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    bad = 0
    t0 = time.time()
    for k, seed in enumerate(range(first, first + count)):
        for fn in (mlp_case, ssim_case):
            try:
                ok, msg = fn(seed)
            except Exception as e:      # noqa: BLE001
                ok, msg = False, "%s raised %r" % (fn.__name__, e)
            if not ok:
                bad += 1
                print("FAIL seed %d %s" % (seed, msg), flush=True)
        if (k + 1) % 100 == 0:
            print("%d seeds, %d failures, %.0f s" % (k + 1, bad, time.time() - t0), flush=True)
    print("done: %d seeds from %d (one MLP and one SSIM case each), %d failures" % (count, first, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
