"""MRI -> acoustic impedance on the GPU: mirror of the reference's `src/impedance.py`
(`ImpedanceEstimator`, :6-53) and of the two helpers it uses from `src/utils.py` (`create_brain_mask` :12-21,
`zscore_normalize` :23-39), over `diffus_mlp_*`, `diffus_brain_mask` and `diffus_masked_stats`.

`ImpedanceEstimator` keeps the reference's module structure (`self.model = nn.Sequential(Linear(1,32), ReLU,
Linear(32,32), ReLU, Linear(32,1))`), so state_dicts are interchangeable; its forward and backward run in the fused
MFMA kernels (hidden activations never reach HBM).  Like the renderer there is no CPU fallback: CPU tensors are moved
to the current HIP device and the result is handed back on the caller's device.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .renderer import _Scope, _device_for, _ptr, _stream, _workspace

AIR_IMPEDANCE = 400.0   # reference src/impedance.py:52
HIDDEN = 32


def _pack(params) -> torch.Tensor:
    return torch.cat([p.reshape(-1) for p in params])


class _MlpFn(torch.autograd.Function):
    """y = out_scale * mlp((x - shift) / div) elementwise over x (any shape); mask (uint8, same shape) optional."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, mask, shift, div, out_scale, fill):
        lib = _lib.load()
        dev = _device_for(x)
        with _Scope(dev):
            xd = x.detach().to(device=dev, dtype=torch.float32).contiguous()
            params = _pack([t.detach().to(device=dev, dtype=torch.float32) for t in (w1, b1, w2, b2, w3, b3)])
            if params.numel() != 1153:
                raise ValueError("the fused MLP is 1 -> 32 -> 32 -> 1 (reference src/impedance.py:10-14)")
            md = mask.to(device=dev, dtype=torch.uint8).contiguous() if mask is not None else None
            y = torch.empty_like(xd)
            if xd.numel():
                _lib.check(lib.diffus_mlp_fwd(_ptr(xd), _ptr(md), xd.numel(), _ptr(params), shift, div, out_scale, fill,
                                              _ptr(y), _stream(dev)), "diffus_mlp_fwd")
        ctx.save_for_backward(xd, params, md)
        ctx.consts = (shift, div, out_scale, x.device, x.dtype, [t.shape for t in (w1, b1, w2, b2, w3, b3)],
                      [t.device for t in (w1, b1, w2, b2, w3, b3)])
        return y.to(x.device)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        xd, params, md = ctx.saved_tensors
        shift, div, out_scale, xdev, xdt, shapes, devs = ctx.consts
        dev = xd.device
        need_x = ctx.needs_input_grad[0]
        with _Scope(dev):
            g = gy.detach().to(device=dev, dtype=torch.float32).contiguous()
            gp = torch.zeros(1153, dtype=torch.float32, device=dev)
            gx = torch.empty_like(xd) if need_x else None
            if xd.numel():
                ws = _workspace(dev, lib.diffus_mlp_workspace_bytes())
                _lib.check(lib.diffus_mlp_bwd(_ptr(xd), _ptr(md), xd.numel(), _ptr(params), shift, div, out_scale, _ptr(g),
                                              _ptr(gp), _ptr(gx), _ptr(ws), ws.numel(), _stream(dev)), "diffus_mlp_bwd")
            elif need_x:
                gx.zero_()
        outs, o = [], 0
        for shp, d, need in zip(shapes, devs, ctx.needs_input_grad[1:7]):
            nel = 1
            for s in shp:
                nel *= s
            outs.append(gp[o:o + nel].reshape(shp).to(d) if need else None)
            o += nel
        return (gx.to(device=xdev, dtype=xdt) if need_x else None, *outs, None, None, None, None, None)


class ImpedanceEstimator(nn.Module):
    """MLP estimating acoustic impedance from normalised intensity (reference src/impedance.py:6-17)."""

    def __init__(self, input_dim: int = 1):
        super().__init__()
        if input_dim != 1:
            raise ValueError("the HIP path implements the reference's input_dim = 1 network")
        self.model = nn.Sequential(
            nn.Linear(input_dim, 32), nn.ReLU(),
            nn.Linear(32, 32), nn.ReLU(),
            nn.Linear(32, 1)
        )

    def _params(self):
        m = self.model
        return m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias

    def forward(self, x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        """x (..., 1) (or any shape: the MLP is applied to every element) -> same shape.  `scale` multiplies the output
        inside the kernel (the notebooks' `model(x) * 1e6` without the extra elementwise pass, forward and backward)."""
        return _MlpFn.apply(x, *self._params(), None, 0.0, 1.0, float(scale), 0.0)

    @classmethod
    def train_model(cls, X: torch.Tensor, y: torch.Tensor, input_dim: int = 1, lr: float = 1e-3,
                    epochs: int = 5000) -> "ImpedanceEstimator":
        """A fresh estimator fitted to the (X, y) pairs: `epochs` full-batch Adam steps on the mean squared error -- the
        recipe of reference src/impedance.py:19-36 --, every forward and backward through the fused MLP kernels."""
        net = cls(input_dim)
        adam = torch.optim.Adam(net.parameters(), lr=lr)
        for _epoch in range(int(epochs)):
            err = torch.nn.functional.mse_loss(net(X), y)
            adam.zero_grad(set_to_none=True)
            err.backward()
            adam.step()
        return net

    @staticmethod
    def compute_impedance_volume(volume: torch.Tensor, model: "ImpedanceEstimator", threshold: float = 50) -> torch.Tensor:
        """Full impedance volume (reference :38-53): brain mask -> z-score inside it -> MLP * 1e6, air = 400 outside.
        Mask, statistics and the masked MLP all run on the GPU; no gradient (the reference wraps it in no_grad)."""
        lib = _lib.load()
        dev = _device_for(volume)
        with torch.no_grad(), _Scope(dev):
            v = volume.detach().to(device=dev, dtype=torch.float32).contiguous()
            mask = create_brain_mask(v, threshold)
            mean, std, _ = masked_stats(v, mask)
            params = _pack([t.detach().to(device=dev, dtype=torch.float32) for t in model._params()])
            out = torch.empty_like(v)
            # (v - mean) / (std + 1e-8) in float32, like zscore_normalize (src/utils.py:38)
            div = (torch.tensor(std, dtype=torch.float32) + 1e-8).item()
            _lib.check(lib.diffus_mlp_fwd(_ptr(v), _ptr(mask.view(torch.uint8)), v.numel(), _ptr(params),
                                          torch.tensor(mean, dtype=torch.float32).item(), div, 1e6, AIR_IMPEDANCE,
                                          _ptr(out), _stream(dev)), "diffus_mlp_fwd")
        return out.to(device=volume.device, dtype=volume.dtype if volume.dtype.is_floating_point else torch.float32)


def create_brain_mask(volume, threshold: float = 50, iterations: int = 2) -> torch.Tensor:
    """mask = volume > threshold, dilated then eroded `iterations` times (6-neighbourhood, outside = 0) -- the
    SciPy calls of reference src/utils.py:18-20.  -> bool tensor on the GPU (CPU for CPU/NumPy input)."""
    lib = _lib.load()
    vt = torch.as_tensor(volume)
    if vt.dim() != 3:
        raise ValueError("create_brain_mask expects a 3-D volume")
    dev = _device_for(vt)
    with _Scope(dev):
        v = vt.detach().to(device=dev, dtype=torch.float32).contiguous()
        d0, d1, d2 = v.shape
        mask = torch.empty(v.shape, dtype=torch.uint8, device=dev)
        ws = _workspace(dev, lib.diffus_brain_mask_workspace_bytes(d0, d1, d2))
        _lib.check(lib.diffus_brain_mask(_ptr(v), d0, d1, d2, float(threshold), int(iterations), _ptr(mask), _ptr(ws),
                                         ws.numel(), _stream(dev)), "diffus_brain_mask")
    return mask.bool().to(vt.device)


def masked_stats(volume: torch.Tensor, mask=None):
    """(mean, unbiased std, count) of the voxels inside the mask, accumulated in float64 on the GPU."""
    lib = _lib.load()
    dev = _device_for(volume)
    with _Scope(dev):
        v = volume.detach().to(device=dev, dtype=torch.float32).contiguous()
        m = mask.to(device=dev).contiguous().view(torch.uint8) if mask is not None else None
        out = torch.empty(3, dtype=torch.float64, device=dev)
        ws = _workspace(dev, lib.diffus_masked_stats_workspace_bytes())
        _lib.check(lib.diffus_masked_stats(_ptr(v), _ptr(m), v.numel(), _ptr(out), _ptr(ws), ws.numel(), _stream(dev)),
                   "diffus_masked_stats")
        mean, std, cnt = out.tolist()
    return mean, std, int(cnt)


def zscore_normalize(volume: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """(volume - mean) / (std + 1e-8) with the statistics of the masked voxels (reference src/utils.py:23-39)."""
    mean, std, _ = masked_stats(volume, mask > 0)
    v = volume.float()
    return (v - torch.tensor(mean, dtype=torch.float32)) / (torch.tensor(std, dtype=torch.float32) + 1e-8)
