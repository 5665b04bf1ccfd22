"""Artifact chain: oracle vs the reference (CPU), HIP vs both (GPU).
Golden G14 = add_speckle_arcs_np -> add_depth_dependent_lateral_blur_np -> sharpen_np run in the
reference with a seeded NumPy RNG; the draws themselves are stored so the deterministic stages can
be compared exactly.  The Philox-seeded path is checked statistically."""
import numpy as np
import pytest
import torch

from conftest import load_golden, maxnorm_rel


def _cases():
    g = load_golden("g14_artifacts")
    return g, [str(t) for t in g["tags"]]


def test_oracle_artifacts_match_reference():
    from oracle import artifacts as oa
    g, tags = _cases()
    for t in tags:
        sr, sl, ms, al = g[f"{t}_params"]
        a1 = oa.speckle(g[f"{t}_frame"], sr, sl, g[f"{t}_radial"], g[f"{t}_local"])
        np.testing.assert_allclose(a1, g[f"{t}_speckle"], rtol=1e-14, atol=0)
        a2 = oa.lateral_blur(g[f"{t}_speckle"], ms)
        assert maxnorm_rel(a2, g[f"{t}_blur"]) < 1e-13, t
        a3 = oa.sharpen(g[f"{t}_blur"], al)
        assert maxnorm_rel(a3, g[f"{t}_sharp"]) < 1e-13, t
        full = oa.chain(g[f"{t}_frame"], sr, sl, ms, al, g[f"{t}_radial"], g[f"{t}_local"])
        assert maxnorm_rel(full, g[f"{t}_full"]) < 1e-12, t        # == plot_beam_frame(artifacts=True) itself
        assert str(g[f"{t}_full_dtype"]) == "torch.float64"


@pytest.mark.gpu
def test_hip_artifacts_match_reference():
    import diffus_amd
    g, tags = _cases()
    for t in tags:
        sr, sl, ms, al = (float(v) for v in g[f"{t}_params"])
        f = torch.from_numpy(g[f"{t}_frame"]).cuda()
        out = diffus_amd.apply_artifacts(f, sr, sl, ms, al, noise=(g[f"{t}_radial"], g[f"{t}_local"]))
        assert out.dtype == torch.float64 and out.shape == f.shape and out.device == f.device
        assert maxnorm_rel(out.cpu().numpy(), g[f"{t}_full"]) < 1e-6, t     # frame is f32; chain in f64


@pytest.mark.gpu
def test_hip_artifacts_seeded_noise_statistics_and_plot_beam_frame():
    import diffus_amd
    from diffus_amd.phantom import phantom, pose_ring
    # noise only: a constant frame isolates the multiplicative factors (max_sigma tiny, alpha 0 => identity filters)
    R, N = 256, 400
    f = torch.ones((R, N), device="cuda")
    a = diffus_amd.apply_artifacts(f, 0.1, 0.05, 1e-6, 0.0, seed=7)
    b = diffus_amd.apply_artifacts(f, 0.1, 0.05, 1e-6, 0.0, seed=7)
    c = diffus_amd.apply_artifacts(f, 0.1, 0.05, 1e-6, 0.0, seed=8)
    assert torch.equal(a, b) and not torch.equal(a, c)                      # reproducible, seed-dependent
    x = a.cpu().numpy()
    depth = np.linspace(0, 1, N)
    radial = x.mean(axis=0)                                                  # ~ radial factor per depth
    local = x / radial[None, :]
    assert abs(radial.mean() - 1) < 0.02 and abs(local.mean() - 1) < 1e-3
    sl = 0.05 * (1 + depth ** 1.5)
    est = local.std(axis=0)
    assert np.all(np.abs(est / sl - 1) < 0.2)                                # std of the local grain follows depth
    sr = 0.1 * (1 + depth ** 2)
    z = (radial - 1) / sr
    assert 0.8 < z.std() < 1.2 and abs(z.mean()) < 0.2                       # radial arcs ~ N(1, sr(depth))
    # through the renderer: float64, right shape, seedable
    vol = torch.from_numpy(phantom(64)).cuda()
    s, d = pose_ring(64, 2, 32)
    Rr = diffus_amd.UltrasoundRenderer(80, 1e-3)
    x1, y1, z1, o1 = Rr.plot_beam_frame(vol, torch.from_numpy(s[0]), torch.from_numpy(d[0]), artifacts=True, start=10, seed=3)
    _, _, _, o2 = Rr.plot_beam_frame(vol, torch.from_numpy(s[0]), torch.from_numpy(d[0]), artifacts=True, start=10, seed=3)
    assert o1.dtype == torch.float64 and o1.shape == (32, 70) and x1.shape == (32, 70) and torch.equal(o1, o2)
    assert torch.isfinite(o1).all() and (o1 >= 0).all()
    with pytest.raises(ZeroDivisionError):
        diffus_amd.apply_artifacts(f, 0.1, 0.05, 0.0, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("R,N,max_sigma", [(40, 70, 3.0), (37, 150, 7.6), (90, 65, 12.0), (5, 33, 4.0), (300, 700, 4.0)])
def test_hip_artifacts_vs_oracle_shapes_and_radii(R, N, max_sigma):
    """Tile borders, rays fewer than the blur radius (repeated reflection), and both blur paths: weights in LDS
    (radius <= 30) and the direct form beyond it."""
    import diffus_amd
    from oracle import artifacts as oa
    rng = np.random.default_rng(R * 1000 + N)
    f = np.abs(rng.normal(0.0, 1.0, size=(2, R, N))).astype(np.float32)
    rs, ls = oa.noise_scales(N, 0.05, 0.1)
    radial = rng.normal(1.0, rs, size=(2, N))
    local = rng.normal(1.0, ls[None, None, :], size=(2, R, N))
    out = diffus_amd.apply_artifacts(torch.from_numpy(f).cuda(), 0.05, 0.1, max_sigma, 2.5, noise=(radial, local)).cpu().numpy()
    for p in range(2):
        ref = oa.chain(f[p], 0.05, 0.1, max_sigma, 2.5, radial[p], local[p])
        assert maxnorm_rel(out[p], ref) < 1e-12, (p, R, N, max_sigma)
