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
from .renderer import _Scope, _as, _device_for, _ptr, _stream, _workspace

log = logging.getLogger("diffus_amd")


def plot_axes(x, y, z):
    """The image plane = the two coordinate axes along which the samples spread most (largest variance first; equal
    variances keep the axis order) -- the choice the reference makes at src/renderer.py:702-707 --, as Python ints.  This
    is the HOST form (one sync for the three variances instead of the reference's three); differentiable_splat itself
    decides on the device (select_axes) and never syncs."""
    import numpy as np
    spread = torch.stack([c.float().var() for c in (x, y, z)]).tolist()
    first, second = np.argsort(-np.asarray(spread), kind="stable")[:2]
    return int(first), int(second)


_COORD_DT = {torch.float32: _lib.DIFFUS_F32, torch.float64: _lib.DIFFUS_F64, torch.int64: _lib.DIFFUS_I64}


def select_axes(x, y, z, dev=None):
    """Device-side axis choice of differentiable_splat (reference src/renderer.py:702-710; diffus_splat_axes): the three
    coordinate planes (any mix of float32 / float64 / int64; other dtypes are cast to float32 first) ->
    (coords (2, n) float32 = the two planes of largest variance, axes (2,) int32 device tensor).  No host sync."""
    lib = _lib.load()
    dev = dev if dev is not None else _device_for(x)
    planes = []
    for c in (x, y, z):
        dt = c.dtype if c.dtype in _COORD_DT else torch.float32
        planes.append(_as(c, dev, dt).reshape(-1))
    n = planes[0].numel()
    if any(p.numel() != n for p in planes):
        raise ValueError("x, y, z must have the same number of elements")
    with _Scope(dev):
        sel = torch.empty((2, n), dtype=torch.float32, device=dev)
        axes = torch.empty(2, dtype=torch.int32, device=dev)
        rc = lib.diffus_splat_axes(_ptr(planes[0]), _COORD_DT[planes[0].dtype], _ptr(planes[1]), _COORD_DT[planes[1].dtype],
                                   _ptr(planes[2]), _COORD_DT[planes[2].dtype], n, _ptr(axes), _ptr(sel[0]), _ptr(sel[1]),
                                   _stream(dev))
    _lib.check(rc, "diffus_splat_axes")
    return sel, axes


class _SplatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c0, c1, val, H, W, sigma, cols):
        lib = _lib.load()
        dev = _device_for(val)
        P, n = val.shape
        with _Scope(dev):
            a = _as(c0, dev, torch.float32)
            b = _as(c1, dev, torch.float32)
            v = _as(val, dev, torch.float32)
            out = torch.empty((P, W, H), dtype=torch.float32, device=dev)
            # the winner raster is kept for the backward when one can follow (three launches there instead of six)
            keep = torch.empty((P, H * W), dtype=torch.int32, device=dev) if ctx.needs_input_grad[2] else None
            nws = lib.diffus_splat_workspace_bytes(P, H, W)
            ws = _workspace(dev, nws)
            rc = lib.diffus_splat_fwd(_ptr(a), _ptr(b), _ptr(v), P, n, int(cols), H, W, float(sigma), _ptr(out), _ptr(keep),
                                      _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(rc, "diffus_splat_fwd")
        ctx.keep = keep
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
        with _Scope(dev):
            g = _as(gout, dev, torch.float32)
            gval = torch.empty((P, n), dtype=torch.float32, device=dev)
            ws = _workspace(dev, lib.diffus_splat_workspace_bytes(P, H, W))
            rc = lib.diffus_splat_bwd(_ptr(a), _ptr(b), P, n, H, W, sigma, _ptr(g), _ptr(gval), _ptr(ctx.keep), _ptr(ws),
                                      ws.numel(), _stream(dev))
        _lib.check(rc, "diffus_splat_bwd")
        if gval.device != vdev or gval.dtype != vdt:
            gval = gval.to(device=vdev, dtype=vdt)
        return None, None, gval, None, None, None, None


def splat_frames(coord0, coord1, intensities, H=256, W=256, sigma=2.0, cols=0):
    """Batched core: coord0/coord1/intensities (P, n) -> (P, W, H).  `cols` = samples per ray when
    the n samples are rays x steps (lets the kernel privatise per patch of rays; same result)."""
    return _SplatFn.apply(coord0, coord1, intensities, int(H), int(W), float(sigma), int(cols))


def differentiable_splat(x, y, z, intensities, H=256, W=256, sigma=2.0):
    """Scan conversion of one frame (reference src/renderer.py:694-737): the samples are dropped onto the plane spanned
    by the two coordinate axes they vary most along, each at its nearest pixel of a W x H raster (the last sample to
    land on a pixel owns it), and the raster and its hit mask are smoothed with a Gaussian of `sigma` and divided.
    x, y, z, intensities: tensors of one shape; returns the (W, H) float32 image on intensities.device -- what the
    reference returns as `output[0, 0].T` --, differentiable in `intensities`."""
    dev = intensities.device
    cdev = _device_for(intensities)
    sel, _ = select_axes(x, y, z, cdev)              # on the device: no .item() round trips (reference :704)
    cols = intensities.shape[-1] if intensities.dim() == 2 else 0
    # (squeeze, not [0]: indexing's backward is a zero-filled (1, W, H) tensor plus a copy into it -- two launches)
    out = splat_frames(sel[0:1], sel[1:2], intensities.reshape(1, -1), H, W, sigma, cols).squeeze(0)
    return out if out.device == dev else out.to(dev)


def rotate_around_apex(x, z, apex, median):
    """Turn the fan so that its median direction points along +z of the image and put its apex at `apex`: every point
    (x - 128, z) is rotated by the angle between (0, 1) and `median`, then shifted -- reference src/renderer.py:655-692
    (the 128 is the reference's too), one launch on x's device (diffus_rotate_around_apex).
    x, z: 1-D coordinate tensors; apex: (x0, z0); median: (dx, dz) -- tuples / arrays like in the reference, or tensors
    already on the device (then nothing is copied from the host and the call can sit inside a captured hipGraph).
    Coordinates are constants of the fan geometry: like in the reference, no gradient flows through this."""
    lib = _lib.load()
    dev = _device_for(x)
    xs, zs = _as(x, dev, torch.float32).reshape(-1), _as(z, dev, torch.float32).reshape(-1)
    if xs.numel() != zs.numel():
        raise ValueError("x and z must have the same number of elements")
    ap = _as(torch.as_tensor(apex, dtype=torch.float32), dev, torch.float32)
    md = _as(torch.as_tensor(median, dtype=torch.float32), dev, torch.float32)
    n = xs.numel()
    with _Scope(dev):
        out = torch.empty((2, n), dtype=torch.float32, device=dev)
        if n:
            _lib.check(lib.diffus_rotate_around_apex(_ptr(xs), _ptr(zs), n, _ptr(ap), _ptr(md), 128.0, _ptr(out[0]), _ptr(out[1]),
                                                     _stream(dev)), "diffus_rotate_around_apex")
    xo, zo = out[0].reshape(x.shape), out[1].reshape(z.shape)
    return (xo, zo) if x.device == dev else (xo.to(x.device), zo.to(x.device))
