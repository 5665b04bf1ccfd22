#!/usr/bin/env python3
"""Fuzz the scatter's slab path (fans that are rigid motions of planar ones, any orientation) against float64 autograd:
tests/test_tilted_fans.py's `_coplanar_case` generator over many seeds, d/dvolume (bricked gradient), d/dsource, d/ddirections
of a random upstream gradient, both samplers, the volume layout by seed, at the sample points the reference's own arithmetic produces.  A gradient fails at
>= 1e-3 max-norm-relative UNLESS the same algorithm run in float32 by torch is as far off there (within 3x: sources far outside the
volume put every sample on the border and the reflection coefficients become differences of nearly equal numbers); exact ties at
the start-crop median are skipped.  Prints one line per failing seed and a progress line every 100 seeds.

    python tools/fuzz_slab.py [first_seed] [count]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if not os.environ.get("FUZZ_DRY"):
    import diffus_amd as da  # noqa: E402
from oracle import autograd_ref as ar  # noqa: E402
from test_tilted_fans import _coplanar_case  # noqa: E402
from test_hip_random import _case as _random_case, _long_case  # noqa: E402


def maxnorm_rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


PLANAR = bool(os.environ.get("FUZZ_PLANAR"))      # control: the same cases with the fans left in the slice
LONG = bool(os.environ.get("FUZZ_LONG"))            # tests/test_hip_random.py's long rays: 1025 ... 2600 samples, chained launches
RANDOM = bool(os.environ.get("FUZZ_RANDOM")) or LONG        # tests/test_hip_random.py's generator instead: every ray its own direction (the 3-D tile)
CROPPED = bool(os.environ.get("FUZZ_CROPPED"))    # only the start > 0 cases, decoupled from the source's place
SEEDS = [int(x) for x in os.environ.get("FUZZ_SEEDS", "").split(",") if x]
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad, worst, skipped, conditioned = [], 0.0, 0, 0
t0 = time.time()
for k, seed in enumerate(SEEDS or range(first, first + count)):
    if CROPPED and not RANDOM and seed % 4 != 3:
        continue
    # (the generator crops exactly the cases whose source sits beside two faces, where most rays are clamped at once and the
    # start-crop median ties; FUZZ_CROPPED=1 runs only the cropped cases, with the source anywhere)
    if RANDOM:
        vol, src, dirs, S, start, alpha = _long_case(seed) if LONG else _random_case(seed)
        if not LONG:
            S = min(S, 300)
            start = min(start, S - 2)
    else:
        vol, src, dirs, S, start, alpha = _coplanar_case(seed, planar=PLANAR, where=((seed // 4) % 3) if CROPPED else None)
    vol = np.abs(vol) + 1e5
    # where the reference's own arithmetic puts the sample points: float32 multiply + add for float32 poses, float64 (then a
    # cast) when the source is float64 (torch's promotion, src/renderer.py:119-124)
    pts = "f32" if (src.dtype == np.float32 and dirs.dtype == np.float32) else "exact"
    for sampler in ("trilinear", "nearest"):
        def reference(dt):
            v_ = torch.from_numpy(vol).to(dt).requires_grad_(True)
            s_ = torch.from_numpy(src).double().requires_grad_(True)
            d_ = torch.from_numpy(dirs).double().requires_grad_(True)
            # (the body of ar.render, with everything after the sample points in `dt`)
            if pts == "f32":
                p64 = ar.ray_points_f32(s_, d_, S)
            elif dirs.dtype == np.float64:      # float64 directions: the whole march in float64, the sampler's cast at the end (pmode 2)
                exact = ar.ray_points(s_, d_, S)
                p64 = exact.detach().float().double() + (exact - exact.detach())
            else:   # float64 source, float32 directions: k * direction is a float32 product, the sum is float64, the sampler casts
                # the point to float32 (src/renderer.py:119-124, :751; csrc ray_point_f, pmode 1) -- straight-through derivatives
                exact = ar.ray_points(s_, d_, S)
                t32 = torch.arange(S, dtype=torch.float32).view(1, S, 1) * torch.from_numpy(dirs).float().unsqueeze(1)
                p64 = (torch.from_numpy(src).double().view(1, 1, 3) + t32.double()).float().double() + (exact - exact.detach())   # (a float32 source is promoted: exact)
            p64.retain_grad()
            p_ = p64.to(dt)
            imp_ = ar.sample_nearest(v_, p_)[0] if sampler == "nearest" else ar.sample_trilinear(v_, p_)
            e_ = ar.echo_scan(ar.start_crop(ar.reflection(imp_), start))
            f_ = e_ * torch.exp(-alpha * torch.arange(e_.shape[1], dtype=dt))[None, :]
            up_ = torch.randn(f_.shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
            (f_ * up_.to(dt)).sum().backward()
            gz = lambda t_: (t_.grad if t_.grad is not None else torch.zeros_like(t_)).double().numpy()     # (nearest: no pose gradient)
            # d/dsource = sum_k dL/dp_k, d/ddirection = sum_k k dL/dp_k, from the per-sample gradients with grid_sample's border rule
            # applied (zero where p <= 0 or p >= dim - 1: the convention golden G9 pins; torch.clamp passes the gradient AT the bound,
            # and a float32 march does land on 0.0 exactly now and then)
            gp_ = p64.grad.clone() if p64.grad is not None else torch.zeros_like(p64)
            hi_ = torch.tensor([d - 1.0 for d in vol.shape], dtype=torch.float64).view(1, 1, 3)
            gp_[(p64.detach() <= 0) | (p64.detach() >= hi_)] = 0.0
            kk_ = torch.arange(S, dtype=torch.float64).view(1, S, 1)
            if s_.grad is not None:
                s_.grad = gp_.sum((0, 1)).to(s_.grad.dtype)
                d_.grad = (gp_ * kk_).sum(1).to(d_.grad.dtype)
            gp_ = gp_.abs()                                                                # what the pose sums are made of
            terms = (float(gp_.sum()), float((gp_ * torch.arange(S, dtype=torch.float64).view(1, S, 1)).sum(1).max()))
            return f_.detach(), gz(v_), gz(s_), gz(d_), up_, terms
        f64, ref, rs, rd, up, terms = reference(torch.float64)
        if start > 0:      # an exact tie at the per-pose median (rays clamped onto the same border voxels): which ray the
            # median's gradient goes to is torch's choice among equals, and a different one is as good a subgradient
            with torch.no_grad():
                imp = (ar.sample_trilinear if sampler == "trilinear" else (lambda v_, p_: ar.sample_nearest(v_, p_)[0]))(
                    torch.from_numpy(vol).double(), ar.ray_points_f32(torch.from_numpy(src).double(), torch.from_numpy(dirs).double(), S)
                    if pts == "f32" else ar.ray_points(torch.from_numpy(src).double(), torch.from_numpy(dirs).double(), S))
                first_kept = ar.reflection(imp)[:, start]
                med = first_kept.median()
                srt = torch.sort(first_kept)[0]
                mid = (len(srt) - 1) // 2
                gaps = [float(srt[j + 1] - srt[j]) for j in (mid - 1, mid) if 0 <= j < len(srt) - 1]
                # ... or a NEAR tie: a coefficient is dZ / sum Z with |dZ| of a few hundred on Z ~ 1.6e6, good to ~1e-4 in float32 --
                # two rays closer than that at the median are ordered by rounding, here and in torch's own float32 run alike
                if int((first_kept == med).sum()) > 1 or (gaps and min(gaps) < 1e-3 * max(abs(float(med)), 1e-30)):
                    skipped += 1
                    continue
        _, n32, ns32, nd32, _, _ = reference(torch.float32)     # the same algorithm in float32: the noise a float32 evaluation carries here
        if os.environ.get("FUZZ_DRY"):      # the reference side alone (runs without a GPU: a check of this script)
            continue
        v = torch.from_numpy(vol).cuda().requires_grad_(True)
        s = torch.from_numpy(src).cuda().requires_grad_(True)
        d = torch.from_numpy(dirs).cuda().requires_grad_(True)
        f = da.render_poses(v, s, d, S, alpha, start=start, sampler=sampler, layout=("bricked", "paired", "canonical")[seed % 3])[0]
        (f * up.float().cuda()).sum().backward()
        gv = v.grad.cpu().numpy()
        errs, noise = {}, {}
        if not np.all(np.isfinite(gv)):
            errs["nonfinite"] = 1.0
        elif np.max(np.abs(ref)) < 1e-14:
            if np.max(np.abs(gv)) >= 1e-9:
                errs["gvol_should_vanish"] = float(np.max(np.abs(gv)))
        else:
            errs["gvol"], noise["gvol"] = maxnorm_rel(gv, ref), maxnorm_rel(n32, ref)
            if sampler == "trilinear":
                errs["gsrc"], noise["gsrc"] = maxnorm_rel(s.grad.cpu().numpy(), rs), maxnorm_rel(ns32, rs)
                errs["gdirs"], noise["gdirs"] = maxnorm_rel(d.grad.cpu().numpy(), rd), maxnorm_rel(nd32, rd)
        # a pose gradient is a SUM over samples (d/dsource: of dL/dp_k, d/ddirection: of k dL/dp_k) that may cancel to far less than
        # its terms: float32 accumulation is then held to 2e-6 of the sum of their magnitudes, beside 1e-3 of the result
        slack = {"gsrc": 2e-6 * terms[0] / max(np.abs(rs).max(), 1e-30), "gdirs": 2e-6 * terms[1] / max(np.abs(rd).max(), 1e-30)}
        failing = {k_: e for k_, e in errs.items() if e >= max(1e-3 + slack.get(k_, 0.0), 3.0 * noise.get(k_, 0.0))}
        for k_, e in errs.items():
            if k_ not in failing and e >= 1e-3:
                conditioned += 1
            elif k_ not in failing:
                worst = max(worst, e)
        if failing:
            bad.append((seed, sampler, errs))
            print("FAIL seed %d %s dims %s R %d S %d start %d: %s (float32 torch: %s)" % (seed, sampler, vol.shape, dirs.shape[0], S, start, errs, noise), flush=True)
    if (k + 1) % 100 == 0:
        print("%d seeds, %d failures, %d median ties skipped, %d above 1e-3 but within 3x float32 torch's own error, worst below 1e-3: %.2e, %.0f s"
              % (k + 1, len(bad), skipped, conditioned, worst, time.time() - t0), flush=True)
print("done: %d seeds from %d, %d failures, %d median ties skipped, %d ill-conditioned (within 3x float32 torch)" % (count, first, len(bad), skipped, conditioned))
sys.exit(1 if bad else 0)
