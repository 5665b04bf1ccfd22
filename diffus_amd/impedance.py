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


def _uniform_stride(t: torch.Tensor):
    """Elements between consecutive entries of `t` read in row-major order, if that is ONE number (a contiguous tensor: 1;
    `volume[:, :, k]` of a contiguous volume: volume.shape[2]); else None."""
    if t.numel() <= 1:
        return 1
    step = None
    expect = None
    for size, stride in zip(reversed(t.shape), reversed(t.stride())):
        if size == 1:
            continue
        if step is None:
            step, expect = stride, stride * size
        else:
            if stride != expect:
                return None
            expect = stride * size
    return step if step and step > 0 else None


def _flat_view(params, dev):
    """The six parameter tensors as ONE (1153,) float32 tensor WITHOUT a copy, when they already lie back to back in one
    storage on `dev` (ImpedanceEstimator keeps them that way: `_flatten`); else None.  The kernels read one packed
    array; `torch.cat` per forward was a launch of its own in every training iteration."""
    first = params[0]
    if first.device != dev:
        return None
    base, end = first.untyped_storage().data_ptr(), first.data_ptr()
    for t in params:
        if (t.dtype != torch.float32 or t.device != dev or not t.is_contiguous() or t.untyped_storage().data_ptr() != base
                or t.data_ptr() != end):
            return None
        end += 4 * t.numel()
    return first.as_strided((sum(t.numel() for t in params),), (1,), first.storage_offset())


class _MlpFn(torch.autograd.Function):
    """y = out_scale * mlp((x - shift) / div) elementwise over x (any shape); mask (uint8, same shape) optional."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, mask, shift, div, out_scale, fill, out=None):
        lib = _lib.load()
        dev = _device_for(x)
        ystride = 1
        if out is not None:
            ystride = _uniform_stride(out)
            if (out.device != dev or out.dtype != torch.float32 or out.numel() != x.numel() or ystride is None):
                raise ValueError("out= must be a float32 tensor on the HIP device with x.numel() elements a uniform stride "
                                 "apart (contiguous, or one slice of a contiguous volume)")
        with _Scope(dev):
            xd = x.detach().to(device=dev, dtype=torch.float32).contiguous()
            ps = [t.detach() for t in (w1, b1, w2, b2, w3, b3)]
            params = _flat_view(ps, dev)
            if params is None:
                params = _pack([t.to(device=dev, dtype=torch.float32) for t in ps])
            if params.numel() != 1153:
                raise ValueError("the fused MLP is 1 -> 32 -> 32 -> 1 (reference src/impedance.py:10-14)")
            md = mask.to(device=dev, dtype=torch.uint8).contiguous() if mask is not None else None
            y = torch.empty_like(xd) if out is None else out.detach()
            if xd.numel():
                _lib.check(lib.diffus_mlp_fwd(_ptr(xd), _ptr(md), xd.numel(), _ptr(params), shift, div, out_scale, fill,
                                              _ptr(y), ystride, _stream(dev)), "diffus_mlp_fwd")
        ctx.save_for_backward(xd, params, md)
        ctx.consts = (shift, div, out_scale, x.device, x.dtype, [t.shape for t in (w1, b1, w2, b2, w3, b3)],
                      [t.device for t in (w1, b1, w2, b2, w3, b3)])
        if out is not None:
            # the kernel wrote through a raw pointer: tell torch (the version counter is shared with out's base, e.g. the volume
            # this slice belongs to -- what the renderer's cache of converted volume copies and autograd's saved-tensor checks key on)
            torch.autograd.graph.increment_version(y)
            return y                    # a fresh tensor object on out's memory (detach(): no view bookkeeping)
        return y.to(x.device)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        xd, params, md = ctx.saved_tensors
        shift, div, out_scale, xdev, xdt, shapes, devs = ctx.consts
        dev = xd.device
        need_x = ctx.needs_input_grad[0]
        with _Scope(dev):
            g = gy.detach().to(device=dev, dtype=torch.float32)
            gstride = _uniform_stride(g)          # e.g. the slice of d/dvolume the renderer hands back: read in place
            if gstride is None:
                g, gstride = g.contiguous(), 1
            # (mlp_reduce_kernel stores every one of the 1153 sums: nothing to zero first unless there is no sample at all)
            gp = torch.empty(1153, dtype=torch.float32, device=dev) if xd.numel() else torch.zeros(1153, dtype=torch.float32, device=dev)
            gx = torch.empty_like(xd) if need_x else None
            if xd.numel():
                ws = _workspace(dev, lib.diffus_mlp_workspace_bytes())
                _lib.check(lib.diffus_mlp_bwd(_ptr(xd), _ptr(md), xd.numel(), _ptr(params), shift, div, out_scale, _ptr(g),
                                              gstride, _ptr(gp), _ptr(gx), _ptr(ws), ws.numel(), _stream(dev)), "diffus_mlp_bwd")
            elif need_x:
                gx.zero_()
        outs, o = [], 0
        for shp, d, need in zip(shapes, devs, ctx.needs_input_grad[1:7]):
            nel = 1
            for s in shp:
                nel *= s
            outs.append(gp[o:o + nel].reshape(shp).to(d) if need else None)
            o += nel
        return (gx.to(device=xdev, dtype=xdt) if need_x else None, *outs, None, None, None, None, None, None)


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

    def _flatten(self):
        """Re-seat the six parameters as views of one flat buffer (same Parameter objects, names, shapes and values: state
        dicts and optimizers are unaffected), so that the kernels can read them in place.  `Module.to()` gives every
        parameter a storage of its own again; the next forward outside a stream capture re-flattens."""
        ps = self._params()
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in ps])
            o = 0
            for p in ps:
                p.data = flat[o:o + p.numel()].view(p.shape)
                o += p.numel()

    def forward(self, x: torch.Tensor, scale: float = 1.0, out: torch.Tensor = None) -> torch.Tensor:
        """x (..., 1) (or any shape: the MLP is applied to every element) -> same shape.  `scale` multiplies the output
        inside the kernel (the notebooks' `model(x) * 1e6` without the extra elementwise pass, forward and backward).
        `out`: write the result there and return it -- a float32 device tensor of x.numel() elements a uniform stride
        apart, e.g. `CapturedStep.slice_view(2, k)`: the slice of the volume the prediction is for (the notebook's
        `Z_vol[:, :, k] = Z_slice` without the copy)."""
        ps = self._params()
        if (ps[0].is_cuda and ps[0].dtype == torch.float32 and _flat_view([p.detach() for p in ps], ps[0].device) is None
                and not torch.cuda.is_current_stream_capturing()):
            self._flatten()
        return _MlpFn.apply(x, *ps, None, 0.0, 1.0, float(scale), 0.0, out)

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
                                          _ptr(out), 1, _stream(dev)), "diffus_mlp_fwd")
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
