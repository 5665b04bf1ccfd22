"""Cases the round-5 fuzzing campaigns found (tools/fuzz_forward.py, tools/fuzz_one_pass.py; DESIGN facts 44-45), kept as tests."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import maxnorm_rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forward_ray_with_moderate_echo_is_reevaluated(oracle):
    """tools/fuzz_forward.py seed 1641: a ray with |echo| = 4.0 was 6.0e-5 from float64 (the oracle's float32 series: 1e-5) while
    the float64 re-evaluation started at |echo| > 8; it starts at 3 now and the frame is inside the usual 5e-5."""
    import diffus_amd
    from test_hip_random import _case
    vol, src, dirs, S, start, alpha = _case(1641)
    x, y, z, fo = oracle.plot_beam_frame(vol, src, dirs, S, alpha, start, sampler="trilinear")
    assert 3.0 < float(np.max(np.abs(fo) * np.exp(alpha * np.arange(fo.shape[-1]))[None, :])) < 8.0     # the band the old threshold missed
    for layout in ("canonical", "bricked", "paired"):
        f, idx = diffus_amd.render_poses(torch.from_numpy(vol).cuda(), torch.from_numpy(src), torch.from_numpy(dirs), S, alpha,
                                         start=start, sampler="trilinear", return_indices=True, layout=layout)
        np.testing.assert_array_equal(idx[0, 0].cpu().numpy(), x)
        assert maxnorm_rel(f[0].cpu().numpy(), fo) < 5e-5, layout


@pytest.mark.parametrize("case,bar", [(1719, 1.5e-3), (9860, 1e-3)])
def test_canonical_volume_gradient_on_tiny_steps(case, bar):
    """tools/fuzz_one_pass.py seed 55, cases 1719 and 9860: canonical volumes marched in 0.03-0.04-voxel steps (thousands of
    contributions per voxel that cancel).  Through the 3-D tile of the canonical gradient (32-bit fixed point, one scale per patch)
    `render_poses -> backward` was 7.5e-3 / 2.7e-3 from float64; through the bricked scratch (doubles) it is where the other
    layouts are."""
    import diffus_amd as da
    from oracle import autograd_ref as ar
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_one_pass as fz
    rng = np.random.default_rng(55)
    for _ in range(case + 1):
        k = fz.gen_case(rng)
    dims, P, R, S, start, sampler, alpha, vol, tgt, scale, f64 = (k[x] for x in ("dims", "P", "R", "S", "start", "sampler", "alpha", "vol", "tgt", "scale", "f64"))
    assert k["layout"] == "canonical" and P == 1
    src, dirs = k["src"], k["dirs"]
    v = torch.from_numpy(vol).cuda().requires_grad_(True)
    f = da.render_poses(v, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), S, alpha, start=start, sampler=sampler, layout="canonical")
    (scale * ((f - torch.from_numpy(tgt).cuda()) ** 2).sum()).backward()
    v64 = torch.from_numpy(vol).double().requires_grad_(True)
    f64_ = ar.render(v64, torch.from_numpy(src[0]).double(), torch.from_numpy(dirs[0]).double(), S, alpha, start, sampler,
                     points="exact" if f64 else "f32")
    (scale * ((f64_ - torch.from_numpy(tgt[0]).double()) ** 2).sum()).backward()
    assert maxnorm_rel(v.grad.cpu().numpy(), v64.grad.numpy()) < bar


@pytest.mark.parametrize("seed", [344, 485, 1575, 2207, 2573, 2650])
def test_long_rows_with_ill_conditioned_echoes(oracle, seed):
    """tools/fuzz_echo.py: rows of 1024 ... 3000 samples (walked in 1024-sample pieces) with |echo| of 25 ... 656 somewhere were
    2e-4 ... 2e-2 from float64 -- the float64 re-evaluation existed for rows of one piece only.  Such a row is now walked a second time
    with its running product carried from piece to piece in float64."""
    import diffus_amd as da
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_echo
    r, _ = fuzz_echo.gen(seed)
    assert r.shape[1] >= 1024
    e = da.compute_echo_traces(torch.from_numpy(r).cuda())[0].cpu().numpy()
    with np.errstate(all="ignore"):
        ref = oracle.echo_scan(r.astype(np.float64), np.float64)
        o32 = oracle.echo_scan(r, np.float32)
    assert float(np.max(np.abs(ref))) > 20.0
    for i in range(r.shape[0]):
        tol = max(2e-5, 30 * maxnorm_rel(o32[i], ref[i]))
        assert maxnorm_rel(e[i], ref[i]) < tol, (seed, i)


@pytest.mark.parametrize("seed,row", [(2207, 2), (344, 1), (2573, 0)])
def test_long_ill_conditioned_rays_are_walked_again_in_float64(oracle, seed, row):
    """Rays of more than 1024 samples go through the forward kernel in chained launches with float32 carries; those with an
    ill-conditioned echo somewhere (here |echo| of 345 ... 656: an impedance profile built from the coefficient rows tools/fuzz_echo.py
    found) are flagged and walked again from the first sample with the running product carried in float64
    (render_fwd_long_repair_kernel).  The sequential float32 oracle is 2e-3 ... 4e-3 from float64 on these rays; the kernels 5e-5."""
    import diffus_amd as da
    from oracle import autograd_ref as ar
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_echo
    rr = fuzz_echo.gen(seed)[0][row].astype(np.float64)
    Z = np.empty(len(rr) + 1)
    Z[0] = 1.5e6
    for n in range(len(rr)):
        Z[n + 1] = Z[n] * (1 + rr[n]) / (1 - rr[n])
    vol = Z.astype(np.float32).reshape(1, 1, -1)
    S = vol.shape[2]
    assert S > 1024
    src = np.zeros(3, np.float32)
    dirs = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 1.0]], np.float32)
    for sampler in ("nearest", "trilinear"):
        f64 = ar.render(torch.from_numpy(vol).double(), torch.from_numpy(src).double(), torch.from_numpy(dirs).double(), S, 1e-4, 0,
                        sampler, points="f32").numpy()
        assert float(np.max(np.abs(f64) * np.exp(1e-4 * np.arange(S))[None, :])) > 100.0
        x, y, z, fo = oracle.plot_beam_frame(vol, src, dirs, S, 1e-4, 0, sampler=sampler)
        assert maxnorm_rel(fo, f64) > 1e-3                                   # what float32 carries cost on such a ray
        for layout in ("canonical", "bricked", "paired"):
            f = da.render_poses(torch.from_numpy(vol).cuda(), torch.from_numpy(src), torch.from_numpy(dirs), S, 1e-4, sampler=sampler,
                                layout=layout)[0].cpu().numpy()
            assert maxnorm_rel(f, f64) < 5e-5, (sampler, layout, maxnorm_rel(f, f64))
    # ... with a start crop (the per-pose median replaces the first kept coefficient, reference :237-244) and a float64 source
    src64 = np.zeros(3, np.float64)
    f64 = ar.render(torch.from_numpy(vol).double(), torch.from_numpy(src64), torch.from_numpy(dirs).double(), S, 1e-4, 3, "trilinear").numpy()
    f = da.render_poses(torch.from_numpy(vol).cuda(), torch.from_numpy(src64), torch.from_numpy(dirs), S, 1e-4, start=3, sampler="trilinear",
                        layout="paired")[0].cpu().numpy()
    assert maxnorm_rel(f, f64) < 5e-5
