"""Scan conversion: mirror of the reference's differentiable_splat / rotate_around_apex
(src/renderer.py:694-737, :655-692) over the HIP C-ABI (diffus_splat_fwd / _bwd).

Same signature and return value as the reference: `differentiable_splat(x, y, z, intensities,
H=256, W=256, sigma=2.0) -> (W, H) float32` on intensities.device.  Gradients reach
`intensities` exactly as torch autograd routes them through the reference (every sample gets
its pixel's gradient); coordinates are rounded, so -- as in the reference -- they get none.
"""
from __future__ import annotations

import logging

import torch

from . import _lib
from .renderer import _device_for, _ptr, _stream, _workspace

log = logging.getLogger("diffus_amd")


def plot_axes(x, y, z):
    """The image plane = the two coordinate axes along which the samples spread most (largest variance first; equal
    variances keep the axis order) -- the choice the reference makes at src/renderer.py:702-707.  One host sync for the
    three variances instead of three."""
    import numpy as np
    spread = torch.stack([c.float().var() for c in (x, y, z)]).tolist()
    first, second = np.argsort(-np.asarray(spread), kind="stable")[:2]
    return int(first), int(second)


class _SplatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c0, c1, val, H, W, sigma, cols):
        lib = _lib.load()
        dev = _device_for(val)
        P, n = val.shape
        with torch.cuda.device(dev):
            a = c0.detach().to(device=dev, dtype=torch.float32).contiguous()
            b = c1.detach().to(device=dev, dtype=torch.float32).contiguous()
            v = val.detach().to(device=dev, dtype=torch.float32).contiguous()
            out = torch.empty((P, W, H), dtype=torch.float32, device=dev)
            nws = lib.diffus_splat_workspace_bytes(P, H, W)
            ws = _workspace(dev, nws)
            rc = lib.diffus_splat_fwd(_ptr(a), _ptr(b), _ptr(v), P, n, int(cols), H, W, float(sigma), _ptr(out), _ptr(ws),
                                      ws.numel(), _stream(dev))
        _lib.check(rc, "diffus_splat_fwd")
        ctx.save_for_backward(a, b)
        ctx.meta = (H, W, float(sigma), val.device, val.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        H, W, sigma, vdev, vdt = ctx.meta
        dev = a.device
        P, n = a.shape
        with torch.cuda.device(dev):
            g = gout.detach().to(device=dev, dtype=torch.float32).contiguous()
            gval = torch.empty((P, n), dtype=torch.float32, device=dev)
            ws = _workspace(dev, lib.diffus_splat_workspace_bytes(P, H, W))
            rc = lib.diffus_splat_bwd(_ptr(a), _ptr(b), P, n, H, W, sigma, _ptr(g), _ptr(gval), _ptr(ws), ws.numel(),
                                      _stream(dev))
        _lib.check(rc, "diffus_splat_bwd")
        return None, None, gval.to(device=vdev, dtype=vdt), None, None, None, None


def splat_frames(coord0, coord1, intensities, H=256, W=256, sigma=2.0, cols=0):
    """Batched core: coord0/coord1/intensities (P, n) -> (P, W, H).  `cols` = samples per ray when
    the n samples are rays x steps (lets the kernel privatise per patch of rays; same result)."""
    return _SplatFn.apply(coord0, coord1, intensities, int(H), int(W), float(sigma), int(cols))


def differentiable_splat(x, y, z, intensities, H=256, W=256, sigma=2.0):
    """
    Differentiable splatting onto the 2D plane of highest variance (reference src/renderer.py:694).
    - x, y, z: tensors of same shape, coordinates in [0, size-1] for each axis
    - intensities: tensor of same shape
    - H, W: output image height and width (for axis0 and axis1)
    Returns the (W, H) image (the reference returns output[0, 0].T).
    """
    coords = [x, y, z]
    axis0, axis1 = plot_axes(x, y, z)
    dev = intensities.device
    cols = intensities.shape[-1] if intensities.dim() == 2 else 0
    out = splat_frames(coords[axis0].reshape(1, -1), coords[axis1].reshape(1, -1), intensities.reshape(1, -1), H, W, sigma, cols)
    return out[0].to(dev)


def rotate_around_apex(x, z, apex, median):
    """Turn the fan so that its median direction points along +z of the image and put its apex at `apex`: every point
    (x - 128, z) is rotated by the angle between (0, 1) and `median`, then shifted.  Host-side geometry on the sample
    coordinates in plain torch, with the values of reference src/renderer.py:655-692 (the 128 is the reference's too).
    x, z: 1-D coordinate tensors; apex: (x0, z0); median: (dx, dz)."""
    heading = torch.as_tensor(median, dtype=torch.float32, device=x.device)
    heading = heading / heading.norm()
    turn = torch.atan2(heading[0], heading[1])
    c, s = torch.cos(turn), torch.sin(turn)
    moved = torch.stack((torch.stack((c, -s)), torch.stack((s, c)))) @ torch.stack((x - 128, z))
    return moved[0] + apex[0], moved[1] + apex[1]
