"""The module functions and mirror methods `from diffus_amd import *` offers beside plot_beam_frame -- prop_single_ray,
propagate_full_rays_batched, custom_nearest_sampler, UltrasoundRenderer.trace_ray / simulate_rays -- against golden G16,
produced by running the reference's own functions (tests/golden/make_golden.py).  CPU part: the oracle's restatements
against the same golden; GPU part: the HIP path."""
import numpy as np
import pytest
import torch

from conftest import load_golden, maxnorm_rel


@pytest.fixture(scope="module")
def g16():
    return load_golden("g16_api_functions")


# ---------------------------------------------------------------- oracle vs the reference's outputs (CPU)
def test_oracle_dense_solution_matches_reference(g16):
    from oracle import dense
    r = torch.from_numpy(g16["r"])
    for dt, tag, tol in ((torch.float32, "f32", 2e-5), (torch.float64, "f64", 1e-12)):
        w = dense.solve_truncated(r.to(dt)).numpy()
        c = dense.propagate_dense(r.to(dt)).numpy()
        for b in range(r.shape[0]):
            assert maxnorm_rel(w[b], g16["w_" + tag][b]) <= tol, (tag, b)
            assert maxnorm_rel(c[b], g16["cum_" + tag][b]) <= tol, (tag, b)
    assert np.all(g16["w_f32"][2] == 0)                       # the ray with a NaN coefficient: all zeros


def test_oracle_point_sampler_matches_reference(g16, oracle):
    x, y, z, v = oracle.sample_points_nearest(g16["vol"], g16["pts"])
    np.testing.assert_array_equal(x, g16["sx"])
    np.testing.assert_array_equal(y, g16["sy"])
    np.testing.assert_array_equal(z, g16["sz"])
    np.testing.assert_array_equal(v, g16["sv"])


# ---------------------------------------------------------------- HIP path (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("tag,tol", [("f32", 2e-5), ("f64", 1e-11)])
def test_prop_single_ray_closed_form_vs_reference_dense_solve(g16, tag, tol):
    import diffus_amd as da
    dt = torch.float32 if tag == "f32" else torch.float64
    r = torch.from_numpy(g16["r"]).to(dt)
    w = da.prop_single_ray(r)
    assert w.dtype == dt and w.shape == (7, 48) and w.device == r.device
    w = w.numpy()
    for b in range(7):
        assert maxnorm_rel(w[b], g16["w_" + tag][b]) <= tol, b
    assert np.all(w[2] == 0)                                  # NaN coefficient -> zeros (linalg.solve + nan_to_num)
    assert np.all(w[4, 0::2] == 1) and np.all(w[4, 1::2] == 0)  # no interfaces reflect: g = 1, d = 0
    np.testing.assert_array_equal(da.prop_single_ray(torch.zeros(3, 0)).numpy(), g16["w_empty"])
    with pytest.raises(ValueError):
        da.prop_single_ray(torch.zeros(5))


@pytest.mark.gpu
def test_propagate_full_rays_batched_vs_reference(g16):
    import diffus_amd as da
    r = torch.from_numpy(g16["r"])
    c = da.propagate_full_rays_batched(r)
    assert c.shape == (7, 24) and c.dtype == torch.float32
    for b in range(7):
        # the reference's own float32 LU is up to 5e-5 from its float64 result on these rows: 1e-4 (north_star's forward
        # tolerance) against it, 1e-5 against the float64 truth
        assert maxnorm_rel(c[b].numpy(), g16["cum_f32"][b]) <= 1e-4, b
        assert maxnorm_rel(c[b].numpy(), g16["cum_f64"][b]) <= 1e-5, b
    # and it is the running sum of compute_echo_traces' series
    e, _ = da.compute_echo_traces(r)
    assert maxnorm_rel(torch.cumsum(e, 1).numpy(), c.numpy()) <= 1e-6


@pytest.mark.gpu
def test_custom_nearest_sampler_arbitrary_points_exact(g16):
    import diffus_amd as da
    vol, pts = torch.from_numpy(g16["vol"]), torch.from_numpy(g16["pts"])
    for p in (pts, pts.double()):
        x, y, z, v = da.custom_nearest_sampler(vol, p, visualize=False)
        assert x.dtype == torch.int64 and x.shape == (5, 17) and v.dtype == torch.float32
        np.testing.assert_array_equal(x.numpy(), g16["sx"])
        np.testing.assert_array_equal(y.numpy(), g16["sy"])
        np.testing.assert_array_equal(z.numpy(), g16["sz"])
        np.testing.assert_array_equal(v.numpy(), g16["sv"])
    x, y, z, v = da.custom_nearest_sampler(vol.cuda(), pts.cuda())          # default arguments, resident tensors
    assert v.is_cuda and torch.equal(v.cpu(), torch.from_numpy(g16["sv"]))


@pytest.mark.gpu
def test_trace_ray_and_simulate_rays_mirrors_vs_reference(g16):
    import diffus_amd as da
    from diffus_amd.phantom import phantom
    vol = torch.from_numpy(phantom(32))
    s, d = torch.from_numpy(g16["pose_src"]), torch.from_numpy(g16["pose_dirs"])
    x, y, z, v = da.UltrasoundRenderer.trace_ray(vol, s, d, 40, 0)
    for got, key in ((x, "tr_x"), (y, "tr_y"), (z, "tr_z"), (v, "tr_v")):
        np.testing.assert_array_equal(got.numpy(), g16[key])
    with pytest.raises(TypeError):
        da.UltrasoundRenderer.trace_ray(vol, s, d, 40)            # `start` is required, as in the reference (:94)
    rr = da.UltrasoundRenderer(num_samples=40, attenuation_coeff=1e-3)
    x, y, z, r = rr.simulate_rays(vol, s, d, start=7)             # num_samples from the constructor; start has no effect
    for got, key in ((x, "sim_x"), (y, "sim_y"), (z, "sim_z"), (r, "sim_r")):
        np.testing.assert_array_equal(got.numpy(), g16[key])
    z1 = rr.simulate_rays(vol, s, d, num_samples=25, MRI=True)    # MRI=True: just the impedances Z1
    np.testing.assert_array_equal(z1.numpy(), g16["sim_mri"])
    # one ray: R comes back 1-D (the reference's R.squeeze(0))
    _, _, _, r1 = rr.simulate_rays(vol, s, d[:1])
    assert r1.shape == (39,) and torch.equal(r1, r[0])


@pytest.mark.gpu
def test_echo_series_is_differentiable_like_the_reference():
    """compute_echo_traces / propagate_full_rays_batched return tensors WITH a grad_fn (the reference's are
    differentiable through torch.linalg.solve); gradients against the reference's own autograd (golden G18, fp64) and, for
    the row that holds a NaN coefficient -- where the reference's whole gradient row is NaN -- against float64 autograd
    over the valid prefix (non-finite contributions are dropped, like in the fused backward).  prop_single_ray, whose
    full solution vector has no backward here, raises instead of silently returning a constant."""
    import diffus_amd as da
    from oracle import autograd_ref as ar
    g = load_golden("g18_echo_autograd")
    w = torch.from_numpy(g["w"]).cuda()
    fin = [0, 1, 2, 4]
    for name, fn in (("echo", lambda x: da.compute_echo_traces(x)[0]), ("prop", da.propagate_full_rays_batched)):
        for dt, tol in ((torch.float64, 1e-5), (torch.float32, 2e-5)):        # the kernel reads float32 coefficients (row 1: r = 0.9995)
            r = torch.from_numpy(g["r"]).to(dt).cuda().requires_grad_(True)
            y = fn(r)
            assert y.grad_fn is not None and y.dtype == dt
            (y * w.to(dt)).sum().backward()
            got = r.grad.cpu().numpy()
            assert np.all(np.isfinite(got))
            for i in fin:
                assert maxnorm_rel(got[i], g["g_" + name][i]) < tol, (name, dt, i)
                assert maxnorm_rel(y[i].detach().cpu().numpy(), g[name][i]) < 1e-5
            # the NaN row: the echoes before the NaN still carry gradient to the coefficients before it; nothing else does
            k = 11
            rp = torch.from_numpy(g["r"][3:4, :k]).requires_grad_(True)
            e = ar.echo_scan(rp)
            if name == "prop":
                # cum_m for m > k repeats cum_k (later echoes are 0): every later weight lands on the prefix sums too
                wk = torch.from_numpy(g["w"][3:4]).clone()
                e = torch.cumsum(e, 1)
                tail = wk[:, k + 1:].sum()
                ((e * wk[:, :k + 1]).sum() + tail * e[:, k]).backward()
            else:
                (e * torch.from_numpy(g["w"][3:4, :k + 1])).sum().backward()
            assert maxnorm_rel(got[3, :k], rp.grad.numpy()[0]) < tol, (name, dt)
            assert np.all(got[3, k:] == 0)
    with pytest.raises(NotImplementedError):
        da.prop_single_ray(torch.from_numpy(g["r"]).cuda().requires_grad_(True))
    assert da.prop_single_ray(torch.from_numpy(g["r"]).cuda()).shape == (5, 82)      # detached input: as before


def test_rasterize_fan_matches_reference():
    """rasterize_fan (reference src/renderer.py:626-653, `[DEMO] REUBEN DATA 46` cell 11): host-side SciPy griddata onto the
    grid of the samples' own coordinates; golden G20 = the reference's output on 60 scattered fan samples."""
    import diffus_amd
    g = load_golden("g20_rasterize_fan")
    img = diffus_amd.rasterize_fan(g["x"], g["z"], g["v"])
    assert img.shape == g["img"].shape == (60, 60)
    np.testing.assert_allclose(img, g["img"], rtol=0, atol=1e-12)
    img_t = diffus_amd.rasterize_fan(torch.from_numpy(g["x"]), torch.from_numpy(g["z"]), torch.from_numpy(g["v"]), output_shape=(8, 8))
    np.testing.assert_array_equal(img_t, img)            # tensors are accepted; output_shape is ignored, as in the reference
