"""B-mode artifact chain on the GPU: mirror of what plot_beam_frame(artifacts=True) does in the
reference (src/renderer.py:264-273) -- speckle arcs, depth-dependent lateral blur, unsharp mask --
over diffus_artifacts.  Returns float64 like the reference.  The reference draws its speckle from
the unseeded global NumPy RNG; here `seed` makes a frame reproducible, and `noise=(radial, local)`
injects explicit factors (used by the parity tests with the very draws the reference made)."""
from __future__ import annotations

import itertools

import torch

from . import _lib
from .renderer import _Scope, _as, _device_for, _ptr, _stream, _workspace

_auto_seed = itertools.count(0x5EED)


def apply_artifacts(frames: torch.Tensor, std_radial: float = 0.01, std_local: float = 0.15, max_sigma: float = 4.0,
                    alpha: float = 5, seed=None, noise=None) -> torch.Tensor:
    """frames (R,N) or (P,R,N) float32 -> same shape, float64, on frames.device."""
    lib = _lib.load()
    dev = _device_for(frames)
    f = _as(frames, dev, torch.float32)
    shape = f.shape
    if f.dim() == 2:
        f = f.unsqueeze(0)
    P, R, N = f.shape
    if N > 1 and not (max_sigma > 0):
        raise ZeroDivisionError("float division by zero")      # what SciPy raises in the reference for sigma = 0
    rad = loc = None
    if noise is not None:
        rad = torch.as_tensor(noise[0], dtype=torch.float64, device=dev).reshape(P, N).contiguous()
        loc = torch.as_tensor(noise[1], dtype=torch.float64, device=dev).reshape(P, R, N).contiguous()
    if seed is None:
        seed = next(_auto_seed)
    with _Scope(dev):
        out = torch.empty((P, R, N), dtype=torch.float64, device=dev)
        ws = _workspace(dev, lib.diffus_artifacts_workspace_bytes(P, R, N))
        rc = lib.diffus_artifacts(_ptr(f), P, R, N, float(std_radial), float(std_local), float(max_sigma), float(alpha),
                                  _ptr(rad), _ptr(loc), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(out), _ptr(ws), ws.numel(),
                                  _stream(dev))
    _lib.check(rc, "diffus_artifacts")
    out = out.reshape(shape)
    return out if out.device == frames.device else out.to(frames.device)
