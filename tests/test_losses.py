"""diffus_amd.ssim_loss (diffus_ssim_loss_fwd / _bwd: min-max normalisation + 1 - SSIM, reference notebook
`[DEMO] Train MRI to Impedance MLP - GPU` cell 16) against the same loss written as plain torch ops (examples/losses.py:
the published SSIM formula with piq's defaults; parity with piq itself is unpinned -- it is not installed and the
reference holds no golden vector of it)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,normalise", [(256, 256, True), (64, 80, True), (47, 33, False), (11, 11, True)])
def test_ssim_loss_value_and_gradient_vs_torch_ops(H, W, normalise):
    import diffus_amd as da
    from losses import minmax01, ssim
    g = torch.Generator().manual_seed(H * 1000 + W)
    ref = torch.rand(H, W, generator=g).cuda()
    img = (torch.rand(H, W, generator=g) * 3.0 - 0.5)
    img[: H // 3] = 0.0                                  # a block of exact zeros, like outside the fan of a splat image
    if normalise:
        img = img.clamp_min(0.0)                         # ... which then also is the (heavily tied) minimum
    img[H // 2, W // 2] = 5.0                            # a unique maximum
    a = img.cuda().requires_grad_(True)
    b = img.cuda().requires_grad_(True)
    la = da.ssim_loss(a, ref, normalise=normalise)
    xb = minmax01(b) if normalise else b
    lb = 1.0 - ssim(xb[None, None], ref[None, None], data_range=1.0)
    assert la.shape == () and abs(float(la) - float(lb)) <= 2e-6
    (2.5 * la).backward()
    (2.5 * lb).backward()
    den = float(b.grad.abs().max())
    assert den > 0 and float((a.grad - b.grad).abs().max()) <= 2e-4 * den, float((a.grad - b.grad).abs().max()) / den
    # identical images: SSIM = 1, loss = 0
    same = da.ssim_loss(ref, ref, normalise=False)
    assert abs(float(same)) <= 1e-6


@pytest.mark.parametrize("H,W", [(512, 512), (400, 640)])
def test_ssim_loss_large_images_are_average_pooled_like_piq(H, W):
    """min(H, W) >= 384: piq.ssim average-pools by max(1, round(min(H, W) / 256)) first."""
    import diffus_amd as da
    from losses import minmax01, ssim
    g = torch.Generator().manual_seed(H + W)
    ref = torch.rand(H, W, generator=g).cuda()
    img = (torch.rand(H, W, generator=g) * 2.0).cuda()
    a = img.clone().requires_grad_(True)
    b = img.clone().requires_grad_(True)
    la = da.ssim_loss(a, ref)
    lb = 1.0 - ssim(minmax01(b)[None, None], ref[None, None], data_range=1.0)
    assert abs(float(la) - float(lb)) <= 2e-6
    la.backward(); lb.backward()
    den = float(b.grad.abs().max())
    assert float((a.grad - b.grad).abs().max()) <= 2e-4 * den


def test_ssim_loss_and_echo_series_detect_in_place_edits():
    """an input edited in place between forward and backward raises (torch's saved-tensor version check), ADVICE r3"""
    import diffus_amd as da
    img = torch.rand(64, 64).cuda().requires_grad_(True)
    ref = torch.rand(64, 64).cuda()
    work = img * 1.0
    l = da.ssim_loss(work, ref)
    work.add_(1.0)
    with pytest.raises(RuntimeError):
        l.backward()
    r = (torch.rand(3, 20).cuda() - 0.5).requires_grad_(True)
    rw = r * 1.0
    e, _ = da.compute_echo_traces(rw)
    rw.mul_(0.5)
    with pytest.raises(RuntimeError):
        e.sum().backward()


def test_ssim_loss_is_capturable_and_validates():
    import diffus_amd as da
    img = torch.rand(64, 64).cuda().requires_grad_(True)
    ref = torch.rand(64, 64).cuda()
    out = torch.zeros((), device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            img.grad = None
            da.ssim_loss(img, ref).backward()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    want = img.grad.clone()
    g = torch.cuda.CUDAGraph()
    img.grad = None
    with torch.cuda.graph(g):
        l = da.ssim_loss(img, ref)
        l.backward()
        out.copy_(l.detach())
    g.replay()
    torch.cuda.synchronize()
    assert float(out) > 0 and float((img.grad - want).abs().max()) <= 1e-6 * float(want.abs().max())
    with pytest.raises(ValueError):
        da.ssim_loss(img, ref[:32])
    with pytest.raises(ValueError):
        da.ssim_loss(img, ref, win=10)
