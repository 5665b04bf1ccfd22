"""The loss the reference's training notebook attaches to the scan-converted image
(`[DEMO] Train MRI to Impedance MLP - GPU` cell 16, `UltrasoundSynthesisModel.loss`):

    synth = (img - img.min()) / (img.max() - img.min() + 1e-8)
    loss  = 1 - piq.ssim(synth[None, None], real[None, None], data_range=1.0)

as ONE autograd node over the HIP C-ABI (diffus_ssim_loss_fwd / _bwd): two launches forward, four backward, instead of
the ~110 kernels (eight MIOpen convolutions) the same thing costs as torch ops -- what keeps a whole training iteration
inside a 0.2 ms hipGraph.  SSIM as in Wang et al. (IEEE TIP 2004) with piq's defaults; piq itself is a third-party
package that is not installed here, so the formula is restated (examples/losses.py has the plain-torch twin the tests
compare with).
"""
from __future__ import annotations

import torch

from . import _lib
from .renderer import _Scope, _as, _device_for, _ptr, _stream


class _SsimLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, ref, normalise, win, sigma, k1, k2):
        lib = _lib.load()
        dev = _device_for(img)
        a, b = _as(img, dev, torch.float32), _as(ref, dev, torch.float32)
        H, W = a.shape
        with _Scope(dev):
            loss = torch.empty((), dtype=torch.float32, device=dev)
            ws = torch.empty(lib.diffus_ssim_workspace_bytes(H, W, win), dtype=torch.uint8, device=dev)
            rc = lib.diffus_ssim_loss_fwd(_ptr(a), _ptr(b), H, W, int(normalise), win, sigma, k1, k2, _ptr(loss), _ptr(ws),
                                          ws.numel(), _stream(dev))
        _lib.check(rc, "diffus_ssim_loss_fwd")
        # save_for_backward (not a plain attribute): torch then raises if `img` / `ref` are edited in place between the
        # forward and the backward -- the workspace keeps min / max / tie counts of the image as it was
        ctx.save_for_backward(a, b)
        ctx.ws = ws
        ctx.meta = (int(normalise), win, sigma, k1, k2, img.device, img.dtype)
        return loss if loss.device == img.device else loss.to(img.device)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gloss):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        ws = ctx.ws
        normalise, win, sigma, k1, k2, idev, idt = ctx.meta
        dev = a.device
        H, W = a.shape
        g = _as(gloss, dev, torch.float32)
        with _Scope(dev):
            gimg = torch.empty((H, W), dtype=torch.float32, device=dev)
            rc = lib.diffus_ssim_loss_bwd(_ptr(a), _ptr(b), H, W, normalise, win, sigma, k1, k2, _ptr(g), _ptr(gimg),
                                          1, _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(rc, "diffus_ssim_loss_bwd")
        if gimg.device != idev or gimg.dtype != idt:
            gimg = gimg.to(device=idev, dtype=idt)
        return gimg, None, None, None, None, None, None


def ssim_loss(img: torch.Tensor, ref: torch.Tensor, normalise: bool = True, win: int = 11, sigma: float = 1.5,
              k1: float = 0.01, k2: float = 0.03) -> torch.Tensor:
    """1 - SSIM(min-max-normalised img, ref) for one (H, W) image pair, differentiable in `img` (scalar tensor).
    `ref` is used as given (the notebook normalises the real image once, up front).

    piq.ssim first average-pools both images by f = max(1, round(min(H, W) / 256)).  Up to 383 pixels that is the
    identity and the whole loss is the one fused node; for larger images the normalisation and the pooling run as
    torch ops in front of the kernel (pooling is linear, so it commutes with the min-max of the UNPOOLED image)."""
    if img.dim() != 2 or ref.shape != img.shape:
        raise ValueError(f"img and ref must be (H, W) tensors of the same shape; got {tuple(img.shape)} and {tuple(ref.shape)}")
    if win % 2 == 0 or not (1 <= win <= 15):
        raise ValueError("win must be odd and at most 15")
    f = max(1, round(min(img.shape) / 256))
    if f > 1:
        x = img
        if normalise:
            lo, hi = x.min(), x.max()
            x = (x - lo) / (hi - lo + 1e-8)
        x = torch.nn.functional.avg_pool2d(x[None, None], f)[0, 0]
        r = torch.nn.functional.avg_pool2d(ref.to(device=x.device, dtype=x.dtype)[None, None], f)[0, 0]
        return _SsimLossFn.apply(x, r, False, int(win), float(sigma), float(k1), float(k2))
    return _SsimLossFn.apply(img, ref, bool(normalise), int(win), float(sigma), float(k1), float(k2))
