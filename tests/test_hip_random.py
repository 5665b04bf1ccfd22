"""Seeded random sweep of the HIP path against the CPU oracle: odd volume shapes (not multiples
of the brick), arbitrary (non-unit, oblique, zero) directions, sources far outside the volume,
random start crops, f32/f64 poses, both samplers, both layouts; forward for all, backward (vs
float64 autograd) for a subset.  Catches indexing mistakes that structured cases never hit."""
import numpy as np
import pytest
import torch

from conftest import maxnorm_rel

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    dims = tuple(int(v) for v in rng.integers(1, 41, size=3))
    if seed % 5 == 0:
        dims = (int(rng.integers(2, 9)), int(rng.integers(30, 70)), int(rng.integers(2, 6)))
    vol = (1.5e6 + 2e5 * rng.standard_normal(dims)).astype(np.float32)
    if seed % 4 == 1:
        vol[rng.random(dims) < 0.2] = 400.0            # air pockets: |r| close to 1
    R = int(rng.integers(1, 9))
    S = int(rng.choice([2, 3, 5, 17, 64, 65, 129, 200, 300, 513]))
    start = int(rng.integers(0, max(1, S - 1))) if (seed % 3 == 0 and S > 2) else 0
    start = min(start, S - 2)
    centre = np.array(dims) / 2
    src = centre + rng.normal(0, 1, 3) * np.array(dims) * (2.0 if seed % 7 == 0 else 0.4)
    dirs = rng.normal(0, 1, (R, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    if seed % 6 == 2:
        dirs *= rng.uniform(0.3, 2.5, (R, 1))          # non-unit steps
    if seed % 11 == 3:
        dirs[0] = 0.0                                   # a ray that never moves
    f64 = seed % 4 == 2
    dt = np.float64 if f64 else np.float32
    return vol, src.astype(dt), dirs.astype(np.float32 if seed % 8 != 6 else dt), S, start, float(10 ** rng.uniform(-4, -1))


@pytest.mark.parametrize("seed", range(48))
def test_random_forward_vs_oracle(oracle, seed):
    import diffus_amd
    vol, src, dirs, S, start, alpha = _case(seed)
    for sampler in ("nearest", "trilinear"):
        x, y, z, fo = oracle.plot_beam_frame(vol, src, dirs, S, alpha, start, sampler=sampler)
        for layout in ("canonical", "bricked", "paired"):
            f, idx = diffus_amd.render_poses(torch.from_numpy(vol).cuda(), torch.from_numpy(src), torch.from_numpy(dirs), S,
                                             alpha, start=start, sampler=sampler, return_indices=True, layout=layout)
            f = f[0].cpu().numpy()
            np.testing.assert_array_equal(idx[0, 0].cpu().numpy(), x)
            np.testing.assert_array_equal(idx[1, 0].cpu().numpy(), y)
            np.testing.assert_array_equal(idx[2, 0].cpu().numpy(), z)
            assert np.all(np.isfinite(f)) == np.all(np.isfinite(fo))
            if np.all(np.isfinite(fo)) and np.max(np.abs(fo)) > 0:
                assert maxnorm_rel(f, fo) < 5e-5, (seed, sampler, layout)
            else:
                np.testing.assert_allclose(f, fo, atol=1e-6)


@pytest.mark.parametrize("seed", range(0, 48, 3))
def test_random_backward_vs_autograd(seed):
    import diffus_amd
    from oracle import autograd_ref as ar
    vol, src, dirs, S, start, alpha = _case(seed)
    if S > 300:
        S = 300
        start = min(start, S - 2)
    vol = np.abs(vol) + 1e5                              # keep r well-conditioned for the fp32-vs-fp64 gradient check
    for sampler in ("nearest", "trilinear"):
        v64 = torch.from_numpy(vol).double().requires_grad_(True)
        s64 = torch.from_numpy(src).double().requires_grad_(True)
        d64 = torch.from_numpy(dirs).double().requires_grad_(True)
        f64 = ar.render(v64, s64, d64, S, alpha, start, sampler)
        up = torch.randn(f64.shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
        (f64 * up).sum().backward()
        for layout in ("canonical", "bricked", "paired"):
            v = torch.from_numpy(vol).cuda().requires_grad_(True)
            s = torch.from_numpy(src).cuda().requires_grad_(True)
            d = torch.from_numpy(dirs).cuda().requires_grad_(True)
            f = diffus_amd.render_poses(v, s, d, S, alpha, start=start, sampler=sampler, layout=layout)[0]
            (f * up.float().cuda()).sum().backward()
            gv = v.grad.cpu().numpy()
            assert np.all(np.isfinite(gv))
            ref = v64.grad.numpy()
            if np.max(np.abs(ref)) < 1e-14:          # degenerate pose (ray inside one or two voxels): everything cancels
                assert np.max(np.abs(gv)) < 1e-9, (seed, sampler, layout)
                continue
            assert maxnorm_rel(gv, ref) < 1e-3, (seed, sampler, layout)
            if sampler == "trilinear":
                assert maxnorm_rel(s.grad.cpu().numpy(), s64.grad.numpy()) < 1e-3, (seed, layout)
                assert maxnorm_rel(d.grad.cpu().numpy(), d64.grad.numpy()) < 1e-3, (seed, layout)


def _long_case(seed):
    rng = np.random.default_rng(1000 + seed)
    dims = tuple(int(v) for v in rng.integers(20, 49, size=3))
    vol = (1.5e6 + 2e5 * rng.standard_normal(dims)).astype(np.float32)
    if seed % 3 == 1:
        vol[rng.random(dims) < 0.1] = 6.4e6
    R = int(rng.integers(1, 5))
    S = int(rng.choice([1025, 1026, 1100, 2047, 2048, 2049, 2600]))
    start = int(rng.integers(1, 40)) if seed % 2 else 0
    centre = np.array(dims) / 2
    src = centre + rng.normal(0, 0.25, 3) * np.array(dims)
    dirs = rng.normal(0, 1, (R, 3))
    dirs *= rng.uniform(0.01, 0.05, (R, 1)) / np.linalg.norm(dirs, axis=1, keepdims=True)   # short steps: stay inside
    dt = np.float64 if seed % 4 == 2 else np.float32
    return vol, src.astype(dt), dirs.astype(np.float32 if seed % 4 != 3 else np.float64), S, start, float(10 ** rng.uniform(-4, -3))


@pytest.mark.parametrize("seed", range(8))
def test_random_long_rays_forward_and_backward(oracle, seed):
    """S - start > 1024 (segmented launches) with random shapes, crops, dtypes; every layout and sampler."""
    import diffus_amd
    from oracle import autograd_ref as ar
    vol, src, dirs, S, start, alpha = _long_case(seed)
    for sampler in ("nearest", "trilinear"):
        x, y, z, fo = oracle.plot_beam_frame(vol, src, dirs, S, alpha, start, sampler=sampler)
        v64 = torch.from_numpy(vol).double().requires_grad_(True)
        s64 = torch.from_numpy(src).double().requires_grad_(True)
        d64 = torch.from_numpy(dirs).double().requires_grad_(True)
        f64 = ar.render(v64, s64, d64, S, alpha, start, sampler)
        up = torch.randn(f64.shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
        (f64 * up).sum().backward()
        for layout in ("canonical", "bricked", "paired"):
            v = torch.from_numpy(vol).cuda().requires_grad_(True)
            s = torch.from_numpy(src).cuda().requires_grad_(True)
            d = torch.from_numpy(dirs).cuda().requires_grad_(True)
            f, idx = diffus_amd.render_poses(v, s, d, S, alpha, start=start, sampler=sampler, return_indices=True, layout=layout)
            np.testing.assert_array_equal(idx[0, 0].cpu().numpy(), x)
            np.testing.assert_array_equal(idx[2, 0].cpu().numpy(), z)
            assert maxnorm_rel(f[0].detach().cpu().numpy(), fo) < 5e-5, (seed, sampler, layout)
            (f[0] * up.float().cuda()).sum().backward()
            ref = v64.grad.numpy()
            assert maxnorm_rel(v.grad.cpu().numpy(), ref) < 1e-3, (seed, sampler, layout)
            if sampler == "trilinear":
                assert maxnorm_rel(s.grad.cpu().numpy(), s64.grad.numpy()) < 1e-3, (seed, layout)
                assert maxnorm_rel(d.grad.cpu().numpy(), d64.grad.numpy()) < 1e-3, (seed, layout)
