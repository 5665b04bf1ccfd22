"""Differentiable torch restatement of the path -- the GRADIENT oracle.

TEST INFRASTRUCTURE ONLY.  The reference has no working backward through
plot_beam_frame, so gradients are checked against torch autograd
over this restatement, run in float64 on the CPU.  Forward semantics are those
of diffus_oracle.c (same citations); everything here is built from
differentiable torch ops so that .backward() yields d/d volume, d/d source and
d/d directions (the last two only for the trilinear sampler: nearest sampling
has integer indices and therefore no pose gradient, exactly as in the
reference, src/renderer.py:754-758).
"""
from __future__ import annotations

import torch


def ray_points(source, directions, S):
    """p[i,k,:] = source + k * dir[i,:]   (src/renderer.py:119-124)."""
    steps = torch.arange(S, dtype=directions.dtype, device=directions.device).view(1, S, 1)
    return source.view(1, 1, 3) + steps * directions.unsqueeze(1)


def sample_nearest(vol, pts):
    """src/renderer.py:754-758.  Indices come from the f32 cast of the points."""
    p32 = pts.detach().float()
    idx = [torch.clamp(p32[..., c].round().long(), 0, vol.shape[c] - 1) for c in range(3)]
    return vol[idx[0], idx[1], idx[2]], idx


def sample_trilinear(vol, pts):
    """grid_sample(bilinear, border, align_corners=True) in voxel coordinates,
    coordinate c <-> dim c; gradient wrt pts is zero outside (0, dim-1)."""
    i0, i1, t = [], [], []
    for c in range(3):
        hi = float(vol.shape[c] - 1)
        pc = pts[..., c].clamp(0.0, hi)            # clamp has zero grad outside
        f = pc.detach().floor()
        i0.append(f.long())
        i1.append(torch.clamp(f.long() + 1, max=vol.shape[c] - 1))
        t.append(pc - f)
    def V(a, b, c):
        return vol[a, b, c]
    c00 = torch.lerp(V(i0[0], i0[1], i0[2]), V(i0[0], i0[1], i1[2]), t[2])
    c01 = torch.lerp(V(i0[0], i1[1], i0[2]), V(i0[0], i1[1], i1[2]), t[2])
    c10 = torch.lerp(V(i1[0], i0[1], i0[2]), V(i1[0], i0[1], i1[2]), t[2])
    c11 = torch.lerp(V(i1[0], i1[1], i0[2]), V(i1[0], i1[1], i1[2]), t[2])
    q0 = torch.lerp(c00, c01, t[1])
    q1 = torch.lerp(c10, c11, t[1])
    return torch.lerp(q0, q1, t[0])


def reflection(imp):
    return (imp[:, 1:] - imp[:, :-1]) / (imp[:, :-1] + imp[:, 1:])


def start_crop(r, start):
    """src/renderer.py:241-244 without the in-place write (autograd-safe)."""
    if start <= 0:
        return r
    r = r[:, start:]
    med = r[:, 0].median()                  # lower median; grad to its source ray
    return torch.cat([med.expand(r.shape[0], 1), r[:, 1:]], dim=1)


def echo_scan(r):
    """echo[:,0]=0, echo[:,n] = (P_n)01/(P_n)11, P_n = M_0...M_{n-1},
    M_k = [[1-2r^2, r],[-r, 1]]  (SURVEY A.3); NaN -> 0 (src/renderer.py:408)."""
    B, N = r.shape
    one = torch.ones(B, dtype=r.dtype, device=r.device)
    zero = torch.zeros_like(one)
    p00, p01, p10, p11 = one, zero, zero, one
    out = [zero]
    for n in range(N):
        x = r[:, n]
        a = 1 - 2 * x * x
        p00, p01, p10, p11 = p00 * a - p01 * x, p00 * x + p01, p10 * a - p11 * x, p10 * x + p11
        # exact power-of-two rescale (constant for differentiation; ratio invariant)
        m = torch.stack([p00, p01, p10, p11]).detach().abs().amax(0)
        ok = torch.isfinite(m) & (m > 0)
        s = torch.where(ok, torch.exp2(-torch.floor(torch.log2(torch.where(ok, m, one)))), one)
        p00, p01, p10, p11 = p00 * s, p01 * s, p10 * s, p11 * s
        v = p01 / p11
        out.append(torch.where(torch.isnan(v), zero, v))
    return torch.stack(out, dim=1)


def ray_points_f32(source, directions, S):
    """The sample points AS THE REFERENCE ROUNDS THEM for float32 poses -- float32 multiply, float32 add
    (src/renderer.py:119-124) -- carried in the dtype of the inputs, with the exact derivatives d p / d source = I,
    d p / d direction = k (straight-through: the value is the rounded point, the gradient that of the formula)."""
    steps32 = torch.arange(S, dtype=torch.float32).view(1, S, 1)
    p32 = source.detach().float().view(1, 1, 3) + steps32 * directions.detach().float().unsqueeze(1)
    exact = ray_points(source, directions, S)
    return p32.to(exact.dtype) + (exact - exact.detach())


def render(vol, source, directions, S, alpha, start=0, sampler="trilinear", points="exact"):
    """Differentiable plot_beam_frame (artifacts=False): -> frame (R, S-start).
    points="f32": sample where the float32 march of the reference lands (ray_points_f32)."""
    pts = ray_points_f32(source, directions, S) if points == "f32" else ray_points(source, directions, S)
    if sampler == "nearest":
        imp, _ = sample_nearest(vol, pts)
    else:
        imp = sample_trilinear(vol, pts)
    r = start_crop(reflection(imp), start)
    echo = echo_scan(r)
    depth = torch.arange(echo.shape[1], dtype=echo.dtype, device=echo.device)
    return echo * torch.exp(-alpha * depth)[None, :]
