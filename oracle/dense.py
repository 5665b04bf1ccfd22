"""Dense-solve form of the echo series -- the reference's own algorithm.

TEST INFRASTRUCTURE ONLY (oracle).  Restates, with torch on the CPU, what
src/renderer.py:367-457 of the reference does: for every truncation depth
n = 0..N build the dense 2(n+1) x 2(n+1) interface system and call
torch.linalg.solve (LAPACK gesv), keep d0, then cumsum and first-difference.
It exists for two reasons:
  * it is the like-for-like CPU baseline timed by bench.py on the GPU box
    (the reference's Python cannot travel there);
  * tests pin the O(N) running-product form (diffus_oracle.c) against it.
Cost is ~(4/3) B N^4 flops, exactly like the reference.
"""
from __future__ import annotations

import torch


def interface_system(r: torch.Tensor):
    """A (B,2n+2,2n+2), b (B,2n+2) for n = r.shape[1] interfaces.

    Unknowns [g0,d0,...,gn,dn].  Rows (src/renderer.py:388-405, with
    t_LR = 1+r, t_RL = 1-r, r_RL = +r from :380-382):
      row 0        : g0 = 1
      row 2k+1     : d_k - r_k g_k - (1-r_k) d_{k+1} = 0
      row 2k+2     : g_{k+1} - (1+r_k) g_k - r_k d_{k+1} = 0
      row 2n+1     : d_n = 0
    """
    B, n = r.shape
    size = 2 * (n + 1)
    A = torch.zeros((B, size, size), dtype=r.dtype, device=r.device)
    b = torch.zeros((B, size), dtype=r.dtype, device=r.device)
    b[:, 0] = 1
    A[:, 0, 0] = 1
    A[:, size - 1, size - 1] = 1
    if n:
        k = torch.arange(n, device=r.device)
        g, d, g1, d1 = 2 * k, 2 * k + 1, 2 * k + 2, 2 * k + 3
        A[:, g1, g] = -(1 + r)
        A[:, g1, d1] = -r
        A[:, g1, g1] = 1
        A[:, d, g] = -r
        A[:, d, d1] = -(1 - r)
        A[:, d, d] = 1
    return A, b


def solve_truncated(r: torch.Tensor) -> torch.Tensor:
    """prop_single_ray: w (B,2n+2) with NaN -> 0 (src/renderer.py:407-408)."""
    A, b = interface_system(r)
    return torch.nan_to_num(torch.linalg.solve(A, b), nan=0.0)


def propagate_dense(r: torch.Tensor) -> torch.Tensor:
    """propagate_full_rays_batched (src/renderer.py:412-436): d0 of every truncated system, cumulated (:435)."""
    B, N = r.shape
    d0 = torch.stack([solve_truncated(r[:, :n])[:, 1] for n in range(N + 1)], dim=1)
    return torch.cumsum(d0, dim=1)


def echo_dense(r: torch.Tensor) -> torch.Tensor:
    """compute_echo_traces (src/renderer.py:412-457): r (B,N) -> echo (B,N+1)."""
    B, N = r.shape
    d0 = torch.stack([solve_truncated(r[:, :n])[:, 1] for n in range(N + 1)], dim=1)
    c = torch.cumsum(d0, dim=1)
    return torch.nn.functional.pad(c[:, 1:] - c[:, :-1], (1, 0))


def plot_beam_frame_dense(vol, source, directions, S, alpha, start=0):
    """Whole reference path (nearest sampler, dense solves) in torch on CPU.

    vol (d0,d1,d2) f32 tensor; returns (x, y, z, frame) like src/renderer.py:275.
    """
    steps = torch.arange(0, S, dtype=torch.float32).view(1, -1, 1)
    pts = (source + steps * directions.unsqueeze(1)).float()        # :119-124, :751
    d0, d1, d2 = vol.shape
    x = torch.clamp(pts[..., 0].round().long(), 0, d0 - 1)          # :754-756
    y = torch.clamp(pts[..., 1].round().long(), 0, d1 - 1)
    z = torch.clamp(pts[..., 2].round().long(), 0, d2 - 1)
    imp = vol[x, y, z]                                              # :758
    r = (imp[:, 1:] - imp[:, :-1]) / (imp[:, :-1] + imp[:, 1:])     # :33,65-68
    if type(start) is float:
        start = int(start * S)
    start = max(0, start)
    if start > 0:                                                   # :241-244
        r = r[:, start:].clone()
        r[:, 0] = r[:, 0].median()
    echo = echo_dense(r)
    depth = torch.arange(echo.shape[1]).float()
    frame = echo * torch.exp(-alpha * depth)[None, :]               # :256-259
    return x[:, start:], y[:, start:], z[:, start:], frame
