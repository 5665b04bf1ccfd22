"""GPU parity tests: the HIP path (through the C-ABI, via diffus_amd) against
  (1) golden vectors produced by the reference itself (tests/golden, G1-G10),
  (2) the CPU oracle on the same seeded inputs,
  (3) torch autograd in float64 over oracle/autograd_ref.py for gradients,
  (4) size-independent properties at BASELINE.json's full sizes.
Tolerances: index planes exact; nearest-mode frame vs reference fp32
<= 1e-4 max-norm-relative per frame; vs fp64 truth <= 1e-5; gradients <= 1e-3
max-norm-relative (float atomics reorder sums).
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, maxnorm_rel
from diffus_amd.phantom import phantom, pose_ring

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def da():
    import diffus_amd
    from diffus_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    assert torch.cuda.is_available()
    return diffus_amd


@pytest.fixture(scope="module")
def vols():
    return {n: phantom(n) for n in (32, 64)}


@pytest.fixture(scope="module")
def vol256():
    return phantom(256)


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ----------------------------------------------------------------------------- stage 2: echo series
def test_echo_traces_golden(da, oracle):
    g1 = load_golden("g1_three_layer")
    e, delays = da.compute_echo_traces(cuda(g1["r"]))
    np.testing.assert_allclose(e.cpu().numpy(), g1["echo"], atol=3e-7)
    assert delays.shape == (3,)
    g2 = load_golden("g2_modeling_choices_phantom")
    e, _ = da.compute_echo_traces(cuda(g2["r"]))
    assert maxnorm_rel(e.cpu().numpy(), g2["echo"]) < 1e-5
    g3 = load_golden("g3_nan")
    e, _ = da.compute_echo_traces(cuda(g3["r"]))
    np.testing.assert_array_equal(e.cpu().numpy(), g3["echo"])
    g4 = load_golden("g4_random_series")
    e = da.compute_echo_traces(cuda(g4["r"]))[0].cpu().numpy()
    orc = oracle.echo_scan(g4["r"])
    for i in range(8):
        assert maxnorm_rel(e[i], g4["echo32"][i]) < 1e-4, i      # the reference in fp32
        assert maxnorm_rel(e[i], g4["echo64"][i]) < 1e-5, i      # the reference in fp64
        assert maxnorm_rel(e[i], orc[i]) < 1e-5, i


def test_echo_traces_grazing_rays_g17(da):
    """The ill-conditioned rays of the benchmark workload (golden G17, the reference's dense solves in fp32 / fp64): the
    HIP echo series is as close to the reference's float64 result as the reference's own float32 is (x4), and at 1e-5
    on the ordinary ray."""
    g = load_golden("g17_grazing_rays")
    e = da.compute_echo_traces(cuda(g["r"]))[0].cpu().numpy()
    for i in range(3):
        ref_noise = maxnorm_rel(g["echo32"][i], g["echo64"][i])
        assert maxnorm_rel(e[i], g["echo64"][i]) < max(1e-5, 4 * ref_noise), i


@pytest.mark.parametrize("N", [0, 1, 2, 63, 64, 127, 128, 255, 300, 511, 1023, 1024, 1500, 2047, 2048, 5000])
def test_echo_traces_sizes(da, oracle, N):
    rng = np.random.default_rng(N)
    r = rng.uniform(-0.6, 0.6, size=(5, N)).astype(np.float32)
    if N > 10:
        r[1, N // 2] = 0.99995  # nearly singular interface
        r[2, 3] = np.nan
        r[3, :] = 0
    e = da.compute_echo_traces(cuda(r))[0].cpu().numpy()
    ref = oracle.echo_scan(r.astype(np.float64), np.float64)
    o32 = oracle.echo_scan(r, np.float32)
    assert e.shape == (5, N + 1)
    assert np.all(e[:, 0] == 0)
    for i in range(5):
        # |r| up to 0.6 at EVERY step is far harsher than tissue: the fp32 running product itself
        # drifts from fp64 (row 1 has a det(M) = 1e-4 interface), so the bar is "the same order as the
        # sequential fp32 oracle's own error" (x10), floor 2e-5
        # (x30 once the row is walked in several 1024-sample pieces: one more rounding per sample)
        tol = max(2e-5, (10 if N < 1024 else 30) * maxnorm_rel(o32[i], ref[i]))
        assert maxnorm_rel(e[i], ref[i]) < tol, (N, i)
    if N > 10:
        assert np.all(e[2, 4:] == 0)


@pytest.mark.parametrize("n", [1024, 3000])
def test_echo_adversarial_growth(da, oracle, n):
    # alternating air/tissue: |P| doubles every two steps (2^500 over 1024 samples) -- renormalisation path
    z = np.where(np.arange(n) % 2 == 0, 400.0, 1.6e6).astype(np.float32)[None, :]
    r = oracle.reflection(z)
    e = da.compute_echo_traces(cuda(r))[0].cpu().numpy()
    ref = oracle.echo_scan(r.astype(np.float64), np.float64)
    assert np.all(np.isfinite(e))
    assert maxnorm_rel(e, ref) < 1e-4


def test_echo_extreme_coefficients(da, oracle):
    """what the scans' power-of-two renormalisation (mat_renorm: the scale from the exponent field) does at the ends
    of its range.  A product that is exactly the zero matrix (r = -1 twice: M(-1)^2 = 0; echo 0/0 -> 0 like the reference's
    nan_to_num, :408) and coefficients of 1e18 (1 - 2 r^2 = -2e36: entries near the top of float32 between rescales) follow the
    float64 series; so do negative-impedance interfaces (|r| of a few 1e9).  (Above |r| ~ 6e18 a matrix entry reaches 2^126
    between two rescales and the scale 2^(127 - e) leaves the normal range: out of reach for impedances -- it needs Z1 + Z2 to
    cancel to 1e-19 of their size -- and a factor 2 below where float32 itself overflows, 1.3e19.)"""
    cases = [np.array([[-1, -1, 0.3, 0.2, -0.5, 0.1]], np.float32),
             np.array([[1e18, -1e18, 1e18, 0.5, -0.3, 0.2]], np.float32),
             np.array([[3e9, 2e9, -4e9, 1e9, 0.5, -0.3, 0.2, 0.1]], np.float32)]
    for r in cases:
        e = da.compute_echo_traces(cuda(r))[0].cpu().numpy()
        with np.errstate(all="ignore"):
            ref = oracle.echo_scan(r.astype(np.float64), np.float64)
        assert np.all(np.isfinite(e))
        assert maxnorm_rel(e, ref) < 1e-5, r


# ----------------------------------------------------------------------------- stage 1: sampling
@pytest.mark.parametrize("layout", ["canonical", "bricked", "paired"])
@pytest.mark.parametrize("sampler", ["nearest", "trilinear"])
def test_trace_rays_vs_oracle(da, oracle, vols, sampler, layout):
    g = load_golden("g5_small_frames")
    for t in [str(x) for x in g["tags"]]:
        n, S = int(g[f"{t}_n"]), int(g[f"{t}_S"])
        src, dirs = g[f"{t}_source"], g[f"{t}_directions"]
        out = da.trace_rays(cuda(vols[n]), torch.from_numpy(src), torch.from_numpy(dirs), S, sampler, layout=layout)
        ix, iy, iz, imp_n = oracle.sample_nearest(vols[n], src, dirs, S)
        idx = out["idx"][:, 0].cpu().numpy()
        np.testing.assert_array_equal(idx[0], ix, err_msg=t)
        np.testing.assert_array_equal(idx[1], iy, err_msg=t)
        np.testing.assert_array_equal(idx[2], iz, err_msg=t)
        imp = imp_n if sampler == "nearest" else oracle.sample_trilinear(vols[n], src, dirs, S)
        np.testing.assert_array_equal(out["imp"][0].cpu().numpy(), imp, err_msg=t)     # same fp32 op sequence
        np.testing.assert_array_equal(out["refl"][0].cpu().numpy(), oracle.reflection(imp), err_msg=t)


def test_trilinear_vs_grid_sample_golden(da):
    g = load_golden("g9_trilinear")
    n, S, alpha = int(g["n"]), int(g["S"]), float(g["alpha"])
    vol = cuda(phantom(n))
    out = da.trace_rays(vol, torch.from_numpy(g["source"]), torch.from_numpy(g["directions"]), S, "trilinear")
    assert maxnorm_rel(out["imp"][0].cpu().numpy(), g["imp"]) < 1e-6
    R = da.UltrasoundRenderer(S, alpha)
    _, _, _, f = R.plot_beam_frame(vol, torch.from_numpy(g["source"]), torch.from_numpy(g["directions"]),
                                   sampler="trilinear")
    assert maxnorm_rel(f.cpu().numpy(), g["frame"]) < 1e-4


# ----------------------------------------------------------------------------- whole frames vs the reference
@pytest.mark.parametrize("layout", ["canonical", "bricked", "paired"])
def test_plot_beam_frame_golden_small(da, oracle, vols, layout):
    g = load_golden("g5_small_frames")
    for t in [str(x) for x in g["tags"]]:
        n, S, alpha, start = int(g[f"{t}_n"]), int(g[f"{t}_S"]), float(g[f"{t}_alpha"]), int(g[f"{t}_start"])
        R = da.UltrasoundRenderer(S, alpha)
        vol = cuda(vols[n])
        x, y, z, f = R.plot_beam_frame(vol, torch.from_numpy(g[f"{t}_source"]), torch.from_numpy(g[f"{t}_directions"]),
                                       plot=False, start=start, layout=layout)
        assert x.dtype == torch.int64 and f.dtype == torch.float32 and f.device == vol.device
        np.testing.assert_array_equal(x.cpu().numpy(), g[f"{t}_x"], err_msg=t)
        np.testing.assert_array_equal(y.cpu().numpy(), g[f"{t}_y"], err_msg=t)
        np.testing.assert_array_equal(z.cpu().numpy(), g[f"{t}_z"], err_msg=t)
        f = f.cpu().numpy()
        assert f.shape == g[f"{t}_frame"].shape
        assert maxnorm_rel(f, g[f"{t}_frame"]) < 1e-4, t
        _, _, _, fo = oracle.plot_beam_frame(vols[n], g[f"{t}_source"], g[f"{t}_directions"], S, alpha, start)
        assert maxnorm_rel(f, fo) < 2e-5, t
        assert np.all(f[:, 0] == 0)


def test_cpu_tensors_are_accepted_like_the_reference(da, vols):
    g = load_golden("g5_small_frames")
    t = "a"
    R = da.UltrasoundRenderer(int(g[f"{t}_S"]), float(g[f"{t}_alpha"]))
    x, y, z, f = R.plot_beam_frame(torch.from_numpy(vols[64]), torch.from_numpy(g[f"{t}_source"]),
                                   torch.from_numpy(g[f"{t}_directions"]))
    assert f.device.type == "cpu" and x.device.type == "cpu"
    assert maxnorm_rel(f.numpy(), g[f"{t}_frame"]) < 1e-4


def test_host_poses_through_the_staging_ring(da, oracle, vols):
    """Host-resident poses go up in one packed asynchronous copy through a ring of pinned staging slots.  Same pose again:
    same frame, bit for bit; a pose that differs in one component: its own frame (against the oracle); back to the first:
    the first; the caller's tensor modified in place: seen."""
    n, S = 64, 90
    vol = cuda(vols[n])
    src, dirs = pose_ring(n, 4, 12)
    s0, d0 = torch.from_numpy(src[0].astype(np.float64)), torch.from_numpy(dirs[0])
    s1 = s0.clone(); s1[1] += 0.75
    R = da.UltrasoundRenderer(S, 1e-3)
    fa = R.plot_beam_frame(vol, s0, d0)[3].clone()
    fb = R.plot_beam_frame(vol, s0, d0)[3].clone()
    fc = R.plot_beam_frame(vol, s1, d0)[3].clone()
    fd = R.plot_beam_frame(vol, s0, d0)[3].clone()
    assert torch.equal(fa, fb) and torch.equal(fa, fd) and not torch.equal(fa, fc)
    for f, s in ((fa, s0), (fc, s1)):
        fo = oracle.plot_beam_frame(vols[n], s.numpy(), dirs[0], S, 1e-3, 0)[3]
        assert maxnorm_rel(f.cpu().numpy(), fo) < 2e-5
    s0[1] += 0.75                                      # the caller's tensor modified in place
    assert torch.equal(R.plot_beam_frame(vol, s0, d0)[3], fc)


def test_config1_golden(da, vol256):
    g = load_golden("g6_config1")
    R = da.UltrasoundRenderer(256, 1e-4)
    x, y, z, f = R.plot_beam_frame(cuda(vol256), torch.from_numpy(g["source"]), torch.from_numpy(g["directions"]))
    np.testing.assert_array_equal(x.cpu().numpy(), g["x"])
    np.testing.assert_array_equal(y.cpu().numpy(), g["y"])
    np.testing.assert_array_equal(z.cpu().numpy(), g["z"])
    assert maxnorm_rel(f.cpu().numpy(), g["frame"]) < 1e-4


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "g10_config2_fwd.npz")),
                    reason="G10 not generated")
def test_config2_golden(da, vol256):
    g = load_golden("g10_config2_fwd")
    R = da.UltrasoundRenderer(512, 1e-4)
    _, _, _, f = R.plot_beam_frame(cuda(vol256), torch.from_numpy(g["source"]), torch.from_numpy(g["directions"]),
                                   return_indices=False)
    assert maxnorm_rel(f.cpu().numpy(), g["frame"]) < 1e-4


# ----------------------------------------------------------------------------- edge cases
@pytest.mark.parametrize("S,start,R", [(2, 0, 2), (3, 1, 3), (48, 46, 5), (65, 0, 1), (130, 1, 7), (257, 0, 3),
                                       (513, 0, 2), (1024, 0, 3), (1030, 6, 2), (1025, 0, 2), (2050, 1, 3),
                                       (3100, 10, 2)])
@pytest.mark.parametrize("sampler", ["nearest", "trilinear"])
@pytest.mark.parametrize("layout", ["canonical", "bricked", "paired"])
def test_shapes_and_edges_vs_oracle(da, oracle, vols, S, start, R, sampler, layout):
    src, dirs = pose_ring(64, 3, R)
    vol = vols[64]
    f = da.render_poses(cuda(vol), torch.from_numpy(src), torch.from_numpy(dirs), S, 2e-3, start=start,
                        sampler=sampler, layout=layout).cpu().numpy()
    assert f.shape == (3, R, S - start)
    for p in range(3):
        _, _, _, fo = oracle.plot_beam_frame(vol, src[p], dirs[p], S, 2e-3, start, sampler=sampler)
        assert maxnorm_rel(f[p], fo) < 3e-5, (p, S, start)


def test_too_many_samples_fails_loudly(da, vols):
    src, dirs = pose_ring(64, 1, 4)
    with pytest.raises(da.DiffusError, match="unsupported"):   # > DIFFUS_MAX_SAMPLES * DIFFUS_MAX_SEGMENTS
        da.render_poses(cuda(vols[64]), torch.from_numpy(src), torch.from_numpy(dirs), 65537, 1e-3)
    with pytest.raises(IndexError):
        da.render_poses(cuda(vols[64]), torch.from_numpy(src), torch.from_numpy(dirs), 48, 1e-3, start=47)


def test_zero_impedance_and_degenerate_volume(da, oracle):
    vol = phantom(32).copy()
    vol[10:14, :, :] = 0.0        # Z1+Z2 = 0 -> NaN r -> echoes zero from there on (reference :408)
    src = np.array([2.0, 16.0, 16.0], np.float32)
    dirs = np.array([[1.0, 0.0, 0.0], [0.9, 0.1, 0.0]], np.float32)
    for sampler, layout in (("nearest", "canonical"), ("trilinear", "canonical"), ("nearest", "bricked"),
                            ("trilinear", "bricked"), ("nearest", "paired"), ("trilinear", "paired")):
        f = da.render_poses(cuda(vol), torch.from_numpy(src), torch.from_numpy(dirs), 30, 1e-3,
                            sampler=sampler, layout=layout).cpu().numpy()[0]
        _, _, _, fo = oracle.plot_beam_frame(vol, src, dirs, 30, 1e-3, 0, sampler=sampler)
        assert np.all(np.isfinite(f))
        np.testing.assert_allclose(f, fo, rtol=1e-5, atol=1e-6)
        assert np.all(f[0, 12:] == 0)
        # the one-pass training step forms the same frame in its adjoint-scan kernel; no gradient is NaN
        one = da.CapturedStep(cuda(vol), torch.from_numpy(src[None]).cuda(), torch.from_numpy(dirs[None]).cuda(), 30, 1e-3,
                              sampler, layout=layout, persistent=False)
        one.step()
        torch.cuda.synchronize()
        np.testing.assert_allclose(one.frame[0].cpu().numpy(), fo, rtol=1e-5, atol=1e-6)
        assert torch.isfinite(one.loss).all() and torch.isfinite(one.gvol).all()
        assert torch.isfinite(one.gsrc).all() and torch.isfinite(one.gdirs).all()
    flat = phantom(32)[:, :, :1].copy()   # d2 == 1: the paired dim-2 load must not be used
    _, _, _, fo = oracle.plot_beam_frame(flat, src, dirs, 30, 1e-3, 0, sampler="trilinear")
    for layout in ("canonical", "bricked", "paired"):
        f = da.render_poses(cuda(flat), torch.from_numpy(src), torch.from_numpy(dirs), 30, 1e-3,
                            sampler="trilinear", layout=layout).cpu().numpy()[0]
        assert maxnorm_rel(f, fo) < 2e-5


# ----------------------------------------------------------------------------- gradients
def _autograd_case(vol_np, src, dirs, S, alpha, start, sampler, gseed=0):
    from oracle import autograd_ref as ar
    vol = torch.from_numpy(vol_np).double().requires_grad_(True)
    s = torch.from_numpy(src).double().requires_grad_(True)
    d = torch.from_numpy(dirs).double().requires_grad_(True)
    # float64 arithmetic at the points the float32 march of the reference lands on (the poses of these tests are
    # float32): marching in float64 flips a nearest index here and there, i.e. moves whole contributions by a voxel
    f = ar.render(vol, s, d, S, alpha, start, sampler, points="f32")
    g = torch.Generator().manual_seed(gseed)
    up = torch.randn(f.shape, generator=g, dtype=torch.float64)
    (f * up).sum().backward()
    return f.detach().numpy(), up.float(), vol.grad.numpy(), (s.grad.numpy() if s.grad is not None else None), \
        (d.grad.numpy() if d.grad is not None else None)


@pytest.mark.parametrize("layout", ["canonical", "bricked", "paired"])
@pytest.mark.parametrize("sampler", ["nearest", "trilinear"])
@pytest.mark.parametrize("S,start", [(48, 0), (48, 7), (150, 0), (300, 12), (513, 0), (700, 0), (1024, 0), (1027, 3)])
def test_backward_vs_float64_autograd(da, vols, sampler, S, start, layout):
    n = 64
    src, dirs = pose_ring(n, 4, 6)
    src, dirs = src[1], dirs[1].copy()
    dirs[:, 2] = 0.21                                   # out of plane: all three lerps carry gradient
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    if S > 300:                                         # long rays: short steps keep the samples inside the 64^3 head
        dirs *= np.float32(40.0 / S)                    # (hundreds of clamped samples piling signed terms on one border
    alpha = 3e-3                                        # voxel measure float32 cancellation, not the kernels)
    f_ref, up, gv_ref, gs_ref, gd_ref = _autograd_case(vols[n], src, dirs, S, alpha, start, sampler)
    vol = cuda(vols[n]).requires_grad_(True)
    s = torch.from_numpy(src).cuda().requires_grad_(True)
    d = torch.from_numpy(dirs).cuda().requires_grad_(True)
    f = da.render_poses(vol, s, d, S, alpha, start=start, sampler=sampler, layout=layout)[0]
    assert maxnorm_rel(f.detach().cpu().numpy(), f_ref) < 2e-5
    (f * up.cuda()).sum().backward()
    assert maxnorm_rel(vol.grad.cpu().numpy(), gv_ref) < 1e-3
    if sampler == "trilinear":
        assert maxnorm_rel(s.grad.cpu().numpy(), gs_ref) < 1e-3
        assert maxnorm_rel(d.grad.cpu().numpy(), gd_ref) < 1e-3
    else:
        assert torch.all(s.grad == 0) and torch.all(d.grad == 0)   # integer indices: no pose gradient


@pytest.mark.parametrize("layout", ["canonical", "paired"])
@pytest.mark.parametrize("sampler", ["nearest", "trilinear"])
@pytest.mark.parametrize("S,start", [(2, 0), (3, 0), (48, 0), (48, 7), (65, 0), (130, 1), (300, 12), (513, 0), (700, 30),
                                     (1024, 0), (1027, 3), (1100, 0)])
def test_one_pass_step_vs_float64_autograd(da, vols, sampler, S, start, layout):
    """diffus_render_step_mse: frame, loss = scale * sum((frame - target)^2) and all three gradients out of one call (the
    frame comes out of the adjoint-scan kernel; N1 > 1024 runs forward + fused backward inside the call) against
    float64 autograd through the oracle's restatement of the reference."""
    from oracle import autograd_ref as ar
    n, P, R, alpha, scale = 64, 2, 6, 3e-3, 0.37
    src, dirs = pose_ring(n, 4, R)
    src, dirs = src[1:1 + P], dirs[1:1 + P].copy()
    dirs[..., 2] = 0.21
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    if S > 300:
        dirs *= np.float32(40.0 / S)
    g = torch.Generator().manual_seed(S)
    target = torch.randn(P, R, S - start, generator=g) * 0.05
    vol64 = torch.from_numpy(vols[n]).double().requires_grad_(True)
    s64 = torch.from_numpy(src).double().requires_grad_(True)
    d64 = torch.from_numpy(dirs).double().requires_grad_(True)
    f_ref = torch.stack([ar.render(vol64, s64[p], d64[p], S, alpha, start, sampler, points="f32") for p in range(P)])
    l_ref = scale * ((f_ref - target.double()) ** 2).sum(dim=(1, 2))
    l_ref.sum().backward()
    one = da.CapturedStep(cuda(vols[n]), torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), S, alpha, sampler,
                          start=start, layout=layout, persistent=False, target=target.cuda(), loss_scale=scale)
    assert one.fused_loss and one.one_pass
    one.frame.fill_(float("nan"))
    one.step()
    torch.cuda.synchronize()
    # (a frame of one or two echoes is nothing but differences of neighbouring float32 samples of ~1.6e6: 1e-4)
    assert maxnorm_rel(one.frame.cpu().numpy(), f_ref.detach().numpy()) < (1e-4 if S <= 3 else 2e-5)
    np.testing.assert_allclose(one.loss.cpu().numpy(), l_ref.detach().numpy(), rtol=1e-4)
    assert maxnorm_rel(one.gvol.cpu().numpy(), vol64.grad.numpy()) < 1e-3
    if sampler == "trilinear":
        assert maxnorm_rel(one.gsrc.cpu().numpy(), s64.grad.numpy()) < 1e-3
        assert maxnorm_rel(one.gdirs.cpu().numpy(), d64.grad.numpy()) < 1e-3
    else:
        assert torch.all(one.gsrc == 0) and torch.all(one.gdirs == 0)


@pytest.mark.parametrize("layout", ["canonical", "paired"])
@pytest.mark.parametrize("sampler", ["nearest", "trilinear"])
@pytest.mark.parametrize("S,start", [(1025, 0), (1500, 0), (2100, 5), (3300, 0)])
def test_long_rays_segmented_forward_and_backward(da, oracle, vols, sampler, S, start, layout):
    """S - start > 1024: the ray is processed as 1024-sample segments chained through carries (forward: the
    running product; backward: the adjoint matrix and the boundary term of d/d impedance).  Short steps keep all
    samples inside the head, so every segment carries signal."""
    n = 64
    src, dirs = pose_ring(n, 4, 5)
    src, dirs = src[2] + np.float32([0.37, 0.41, 0.29]), dirs[2].copy()   # off the lattice: the central ray
    dirs[:, 2] = 0.17                                   # would otherwise run along a kink of the interpolant
    dirs *= (0.55 * n / S) / np.linalg.norm(dirs, axis=1, keepdims=True)
    alpha = 8e-4
    _, _, _, fo = oracle.plot_beam_frame(vols[n], src, dirs, S, alpha, start, sampler=sampler)
    f_ref, up, gv_ref, gs_ref, gd_ref = _autograd_case(vols[n], src, dirs, S, alpha, start, sampler)
    vol = cuda(vols[n]).requires_grad_(True)
    s = torch.from_numpy(src).cuda().requires_grad_(True)
    d = torch.from_numpy(dirs).cuda().requires_grad_(True)
    f = da.render_poses(vol, s, d, S, alpha, start=start, sampler=sampler, layout=layout)[0]
    fh = f.detach().cpu().numpy()
    assert fh.shape == (5, S - start)
    assert np.abs(fo[:, 1024:]).max() > 1e-3 * np.abs(fo).max()      # the later segments are not trivially zero
    assert maxnorm_rel(fh, fo) < 3e-5
    assert maxnorm_rel(fh, f_ref) < 3e-5
    (f * up.cuda()).sum().backward()
    assert maxnorm_rel(vol.grad.cpu().numpy(), gv_ref) < 1e-3
    if sampler == "trilinear":
        assert maxnorm_rel(s.grad.cpu().numpy(), gs_ref) < 1e-3
        assert maxnorm_rel(d.grad.cpu().numpy(), gd_ref) < 1e-3


def test_backward_golden_volume_grad(da):
    g = load_golden("g7_volume_grad")
    n, S, alpha = int(g["n"]), int(g["S"]), float(g["alpha"])
    vol = cuda(phantom(n)).requires_grad_(True)
    f = da.render_poses(vol, torch.from_numpy(g["source"]), torch.from_numpy(g["directions"]), S, alpha)[0]
    assert maxnorm_rel(f.detach().cpu().numpy(), g["frame"]) < 1e-4
    (f ** 2).sum().backward()
    gv = vol.grad.flatten().cpu().numpy()
    ref = np.zeros_like(gv)
    ref[g["grad_index"]] = g["grad_value"]
    assert maxnorm_rel(gv, ref) < 1e-3          # the reference's own autograd (fp32 LU backward)


def test_backward_batched_shared_fan_and_f64_pose(da, vols):
    n, S, alpha = 64, 100, 1e-3
    src, dirs = pose_ring(n, 5, 8)
    fan = dirs[0].copy()
    vol = cuda(vols[n]).requires_grad_(True)
    s = torch.from_numpy(src).double().cuda().requires_grad_(True)     # f64 sources, shared f32 fan
    d = torch.from_numpy(fan).cuda().requires_grad_(True)
    f = da.render_poses(vol, s, d, S, alpha, start=3, sampler="trilinear")
    assert f.shape == (5, 8, 97)
    (f ** 2).sum().backward()
    assert s.grad.dtype == torch.float64 and s.grad.shape == (5, 3) and d.grad.shape == (8, 3)
    gv = torch.zeros_like(vol)
    gd = torch.zeros_like(d)
    for p in range(5):
        v2 = cuda(vols[n]).requires_grad_(True)
        s2 = torch.from_numpy(src[p]).double().cuda().requires_grad_(True)
        d2 = torch.from_numpy(fan).cuda().requires_grad_(True)
        f2 = da.render_poses(v2, s2, d2, S, alpha, start=3, sampler="trilinear")
        torch.testing.assert_close(f2[0], f[p], rtol=0, atol=0)
        (f2 ** 2).sum().backward()
        gv += v2.grad
        gd += d2.grad
        assert maxnorm_rel(s2.grad.cpu().numpy(), s.grad[p].cpu().numpy()) < 1e-5
    assert maxnorm_rel(vol.grad.cpu().numpy(), gv.cpu().numpy()) < 1e-4
    assert maxnorm_rel(d.grad.cpu().numpy(), gd.cpu().numpy()) < 1e-4


def test_pose_gradient_finite_difference(da, vols):
    # independent of any autograd: central differences on the HIP forward itself, smooth volume
    n = 64
    u = np.arange(n, dtype=np.float64) / (n - 1)
    vol = (1.6e6 + 2e5 * np.sin(5 * u)[:, None, None] * np.cos(4 * u)[None, :, None] * np.sin(3 * u + 1)[None, None, :])
    vol = cuda(vol.astype(np.float32))
    src0 = torch.tensor([20.3, 22.7, 30.4], dtype=torch.float64)
    d0 = torch.tensor([[0.6, 0.7, 0.3872983], [0.8, 0.5, 0.3316625]], dtype=torch.float64)
    S, alpha = 40, 1e-2
    w = torch.linspace(0.5, 1.5, 2 * S, dtype=torch.float64).reshape(2, S).cuda()

    def loss(s, d):
        return (da.render_poses(vol, s, d, S, alpha, sampler="trilinear")[0].double() * w).sum()

    s = src0.clone().cuda().requires_grad_(True)
    d = d0.clone().cuda().requires_grad_(True)
    loss(s, d).backward()
    h = 1e-2
    for c in range(3):
        e = torch.zeros(3, dtype=torch.float64); e[c] = h
        fd = (loss((src0 + e).cuda(), d0.cuda()) - loss((src0 - e).cuda(), d0.cuda())).item() / (2 * h)
        assert abs(fd - s.grad[c].item()) < 2e-2 * max(1e-12, s.grad.abs().max().item()), (c, fd, s.grad)


# ----------------------------------------------------------------------------- full-size (BASELINE configs 2/3)
def test_full_size_batch_vs_oracle_and_properties(da, oracle, vol256):
    P, R, S, alpha = 32, 256, 512, 1e-4
    src, dirs = pose_ring(256, P, R)
    vol = cuda(vol256)
    s, d = torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda()
    for sampler, layout in (("nearest", "bricked"), ("trilinear", "bricked"), ("trilinear", "canonical"),
                            ("trilinear", "paired"), ("nearest", "paired")):
        f = da.render_poses(vol, s, d, S, alpha, sampler=sampler, layout=layout)
        assert f.shape == (P, R, S)
        fc = f.cpu().numpy()
        assert np.all(np.isfinite(fc)) and np.all(fc[:, :, 0] == 0)
        for p in (0, 7, 19, 31):                       # the O(N) C oracle does a 256x512 frame in milliseconds
            _, _, _, fo = oracle.plot_beam_frame(vol256, src[p], dirs[p], S, alpha, 0, sampler=sampler)
            assert maxnorm_rel(fc[p], fo) < 2e-5, (sampler, p)
        # determinism and batch-independence of the forward
        f2 = da.render_poses(vol, s[5:6], d[5:6], S, alpha, sampler=sampler, layout=layout)
        assert torch.equal(f2[0], f[5])
        assert torch.equal(da.render_poses(vol, s, d, S, alpha, sampler=sampler, layout=layout), f)
        if layout != "canonical":     # the layout must not change a single bit of the frame
            assert torch.equal(da.render_poses(vol, s, d, S, alpha, sampler=sampler, layout="canonical"), f)


def test_full_size_backward_properties(da, vol256):
    P, R, S, alpha = 32, 256, 512, 1e-4
    src, dirs = pose_ring(256, P, R)
    vol = cuda(vol256).requires_grad_(True)
    s = torch.from_numpy(src).cuda().requires_grad_(True)
    d = torch.from_numpy(dirs).cuda().requires_grad_(True)
    f = da.render_poses(vol, s, d, S, alpha, sampler="trilinear")
    g1 = torch.autograd.grad(f, (vol, s, d), grad_outputs=2 * f.detach(), retain_graph=True)
    g2 = torch.autograd.grad(f, (vol, s, d), grad_outputs=6 * f.detach(), retain_graph=True)
    for a, b in zip(g1, g2):                            # backward is linear in the upstream gradient
        assert torch.all(torch.isfinite(a))
        assert maxnorm_rel((3 * a).cpu().numpy(), b.cpu().numpy()) < 1e-4
    # volume gradient only touches voxels on the fans (two dim-2 slabs around each apex plane)
    nzz = torch.nonzero(g1[0].abs().sum((0, 1))).flatten().cpu().numpy()
    zs = src[:, 2]
    assert nzz.min() >= np.floor(zs.min()) and nzz.max() <= np.ceil(zs.max()) + 1
    # zero upstream gradient -> zero gradients
    g0 = torch.autograd.grad(f, (vol, s, d), grad_outputs=torch.zeros_like(f))
    assert all(torch.all(x == 0) for x in g0)


def test_config5_shape_512_volume_vs_oracle_and_backward(da, oracle):
    """BASELINE config 5: a 512^3 volume, 512 rays x 1024 steps (one launch covers exactly 1024 samples), fwd + bwd."""
    n, P, R, S, alpha = 512, 2, 512, 1024, 1e-4
    v = phantom(n, variant=3)
    src, dirs = pose_ring(n, 8, R)
    src, dirs = src[[1, 6]], dirs[[1, 6]]
    vol = cuda(v).requires_grad_(True)
    s = torch.from_numpy(src).cuda().requires_grad_(True)
    d = torch.from_numpy(dirs).cuda().requires_grad_(True)
    f = da.render_poses(vol, s, d, S, alpha, sampler="trilinear")
    fc = f.detach().cpu().numpy()
    assert fc.shape == (P, R, S) and np.all(np.isfinite(fc))
    for p in range(P):
        _, _, _, fo = oracle.plot_beam_frame(v, src[p], dirs[p], S, alpha, 0, sampler="trilinear")
        assert maxnorm_rel(fc[p], fo) < 2e-5, p
    fn = da.render_poses(vol.detach(), s.detach(), d.detach(), S, alpha, sampler="nearest")
    _, _, _, fo = oracle.plot_beam_frame(v, src[0], dirs[0], S, alpha, 0, sampler="nearest")
    assert maxnorm_rel(fn[0].cpu().numpy(), fo) < 2e-5
    del fn
    g1 = torch.autograd.grad(f, (vol, s, d), grad_outputs=2 * f.detach(), retain_graph=True)
    g2 = torch.autograd.grad(f, (vol, s, d), grad_outputs=-4 * f.detach())
    for a, b in zip(g1, g2):                            # linear in the upstream gradient, at full size
        assert torch.all(torch.isfinite(a)) and float(a.abs().max()) > 0
        assert maxnorm_rel((-2 * a).cpu().numpy(), b.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("n,ring,R,S,pose", [(256, 32, 256, 512, 3), (512, 8, 512, 1024, 5)])
def test_full_size_one_pose_gradient_values_vs_float64_autograd(da, n, ring, R, S, pose):
    """VALUES, not just properties, at the BASELINE shapes: one config-2 pose (256 rays x 512 steps, 256^3) and one
    config-5 pose (512 rays x 1024 steps, 512^3): frame, d/dvolume, d/dsource and d/ddirections of sum(frame^2) against
    torch autograd over the float64 restatement (oracle/autograd_ref.py), max-norm-relative <= 1e-3."""
    from oracle import autograd_ref as ar
    alpha = 1e-4
    v = phantom(n, variant=1 if n == 512 else 0)
    src, dirs = pose_ring(n, ring, R)
    vol = cuda(v).requires_grad_(True)
    s = torch.from_numpy(src[pose:pose + 1]).cuda().requires_grad_(True)
    d = torch.from_numpy(dirs[pose:pose + 1]).cuda().requires_grad_(True)
    f = da.render_poses(vol, s, d, S, alpha, sampler="trilinear")
    (f ** 2).sum().backward()
    got = (f.detach()[0].cpu().numpy(), s.grad[0].cpu().numpy(), d.grad[0].cpu().numpy())
    gv = vol.grad.cpu()
    del vol, f
    torch.cuda.empty_cache()
    v64 = torch.from_numpy(v).double().requires_grad_(True)
    s64 = torch.from_numpy(src[pose]).double().requires_grad_(True)
    d64 = torch.from_numpy(dirs[pose]).double().requires_grad_(True)
    # float64 arithmetic at the sample points the float32 march of the reference lands on (:119-124; ray_points_f32):
    # marching in float64 instead moves every point ~1e-5 voxel, next to bone/air steps of 6e6 per voxel -- 1e-4 on the
    # frame and more on d/dsource, a sum of half a million signed terms -- which would measure the poses' dtype, not
    # the kernels
    fr = ar.render(v64, s64, d64, S, alpha, 0, "trilinear", points="f32")
    (fr ** 2).sum().backward()
    assert maxnorm_rel(got[0], fr.detach().numpy()) < 2e-5
    assert maxnorm_rel(got[1], s64.grad.numpy()) < 1e-3
    assert maxnorm_rel(got[2], d64.grad.numpy()) < 1e-3
    gref = v64.grad
    den = float(gref.abs().max())
    err = 0.0
    for i0 in range(0, n, 64):                                     # slab by slab: no GiB-sized temporaries
        err = max(err, float((gv[i0:i0 + 64].double() - gref[i0:i0 + 64]).abs().max()))
    assert den > 0 and err / den < 1e-3, err / den
    assert int((gv != 0).sum()) > 0.5 * int((gref != 0).sum())      # the support is there, not just the peak


@pytest.mark.parametrize("sampler", ["trilinear", "nearest"])
@pytest.mark.parametrize("alpha", [1e-4, 0.5])
def test_fixed_point_scatter_error_bound_per_voxel(da, alpha, sampler):
    """The scatter accumulates a patch (32 rays x 32 steps) in integer fixed point with ONE power-of-two scale per
    patch.  With the reference's default attenuation_coeff = 0.5 the upstream gradient falls by e per step, 2^-46 over a
    patch: a 32-bit accumulator rounds the deep end of every patch to zero -- invisible in the max norm,
    fatal for per-element-normalised optimisers.  Planar patches therefore accumulate in 64 bits (quantum 2^-61 of the
    patch's sum of |zbar|).  The bound is asserted on the scatter ALONE: the kernel's own zbar (read back from the
    workspace between the two backward stages) is scattered exactly, in float64, on the CPU; per voxel, relative to
    the voxel's mass m_v = sum_s |zbar_s| w_sv (float32 atomics cannot beat a rounding error of that):
        |g_v - exact_v| <= 4e-6 m_v + 2^-40 max m."""
    from diffus_amd import CapturedStep, _lib
    from oracle import autograd_ref as ar
    n, R, S = 64, 64, 96
    v = phantom(n)
    src, dirs = pose_ring(n, 8, R)
    hp = CapturedStep(cuda(v), torch.from_numpy(src[2:3]).cuda(), torch.from_numpy(dirs[2:3]).cuda(), S, alpha, sampler,
                      persistent=False)
    gather = ar.sample_trilinear if sampler == "trilinear" else (lambda vol, p: ar.sample_nearest(vol, p)[0])
    hp.fwd(); hp.loss_and_grad(); hp.zero_grad()
    hp.bwd(_lib.BWD_SCAN)
    torch.cuda.synchronize()
    off = _lib.load().diffus_workspace_zbar_offset(1, R, S, 0)
    zbar = hp.ws[off:off + 4 * R * S].view(torch.float32).reshape(R, S).cpu().double()
    hp.bwd(_lib.BWD_SCATTER); hp.finish_grad()
    torch.cuda.synchronize()
    g = hp.gvol.cpu().double()
    pts = ar.ray_points_f32(torch.from_numpy(src[2]).double(), torch.from_numpy(dirs[2]).double(), S)
    ve = torch.from_numpy(v).double().requires_grad_(True)
    (zbar * gather(ve, pts)).sum().backward()
    vm = torch.from_numpy(v).double().requires_grad_(True)
    (zbar.abs() * gather(vm, pts)).sum().backward()
    exact, mass = ve.grad, vm.grad
    tol = 4e-6 * mass + 2.0 ** -40 * float(mass.max())
    worst = float(((g - exact).abs() / tol).max())
    assert worst <= 1.0, worst
    # the deep end is really there: voxels whose mass is 2^-30 .. 2^-38 of the largest carry their gradient
    deep = (mass < 2.0 ** -30 * float(mass.max())) & (mass > 2.0 ** -38 * float(mass.max()))
    if alpha == 0.5:
        assert int(deep.sum()) > (50 if sampler == "trilinear" else 10)
        assert int((g[deep] != 0).sum()) > 0.9 * int(deep.sum())


@pytest.mark.parametrize("sampler", ["trilinear", "nearest"])
@pytest.mark.parametrize("step", [1.0, 1.7, 2.6, 6.0])
def test_planar_scatter_long_steps_every_tile_grouping(da, sampler, step):
    """The planar scatter keeps a patch (32 rays x 32 steps) in one LDS tile when its bounding box fits (unit steps: always),
    splits the patch's waves into 2 or 4 groups when it does not, and adds straight to memory when even one wave's strip
    is too wide.  Step lengths 1.7 / 2.6 / 6 voxels in a wide slice walk through all of these; every one of them against
    float64 autograd of the same render (the reference accepts any direction norm: renderer.py:94-110)."""
    rng = np.random.default_rng(11)
    v = (1.5e6 + 2e5 * rng.standard_normal((320, 320, 4))).astype(np.float32)
    R, S = 64, 96
    ang = np.linspace(0.35, 1.2, R)
    dirs = (step * np.stack([np.cos(ang), np.sin(ang), np.zeros(R)], 1)).astype(np.float32)
    src = np.array([6.3, 9.1, 1.4], np.float32)
    f_ref, up, gv_ref, _, _ = _autograd_case(v, src, dirs, S, 2e-3, 0, sampler)
    for layout in ("paired", "canonical"):
        vol = cuda(v).requires_grad_(True)
        f = da.render_poses(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), S, 2e-3, sampler=sampler,
                            layout=layout)[0]
        assert maxnorm_rel(f.detach().cpu().numpy(), f_ref) < 2e-5
        (f * up.cuda()).sum().backward()
        assert maxnorm_rel(vol.grad.cpu().numpy(), gv_ref) < 1e-3, (layout, step)
        # and the total is conserved to float32 rounding: no contribution lost between groups
        assert abs(float(vol.grad.double().sum()) - gv_ref.sum()) <= 1e-4 * np.abs(gv_ref).sum()


def test_wide_slices_fall_back_to_a_layout_that_fits(da, oracle):
    """Bricked / paired records address a brick row with a 24-bit multiply: slices wider than ~2^17 bricks do not fit.
    `layout="auto"` then stays canonical (same frames), an explicit request is refused with the ABI's error."""
    rng = np.random.default_rng(5)
    v = (1.5e6 + 1e5 * rng.standard_normal((4, 1100, 1100))).astype(np.float32)
    src = np.array([1.5, 20.0, 30.0], np.float32)
    ang = np.linspace(0.2, 1.2, 5)
    dirs = np.stack([np.zeros(5), np.cos(ang), np.sin(ang)], 1).astype(np.float32)
    assert not da.renderer.layout_fits("paired", v.shape) and not da.renderer.layout_fits("bricked", v.shape)
    vol = cuda(v)
    for sampler in ("nearest", "trilinear"):
        f = da.render_poses(vol, torch.from_numpy(src), torch.from_numpy(dirs), 300, 1e-3, sampler=sampler, layout="auto")
        f = da.render_poses(vol, torch.from_numpy(src), torch.from_numpy(dirs), 300, 1e-3, sampler=sampler, layout="auto")
        _, _, _, fo = oracle.plot_beam_frame(v, src, dirs, 300, 1e-3, 0, sampler=sampler)
        assert maxnorm_rel(f[0].cpu().numpy(), fo) < 2e-5
    with pytest.raises(da.DiffusError):
        da.render_poses(vol, torch.from_numpy(src), torch.from_numpy(dirs), 300, 1e-3, sampler="trilinear", layout="paired")


# ----------------------------------------------------------------------------- bricked layout
@pytest.mark.parametrize("shape", [(4, 4, 2), (5, 7, 3), (64, 64, 64), (33, 70, 129), (1, 1, 1), (3, 2, 131)])
def test_brick_roundtrip(da, shape):
    g = torch.Generator().manual_seed(1)
    v = torch.randn(shape, generator=g).cuda()
    b = da.brick_volume(v)
    d0, d1, d2 = shape
    assert b.numel() == ((d0 + 3) // 4) * ((d1 + 3) // 4) * ((d2 + 1) // 2) * 32
    assert torch.equal(da.unbrick_volume(b, shape), v)
    # element placement: brick-major, (x&3, y&3, z&1) inside
    x, y, z = d0 - 1, d1 // 2, d2 - 1
    nb1, nb2 = (d1 + 3) // 4, (d2 + 1) // 2
    off = (((x >> 2) * nb1 + (y >> 2)) * nb2 + (z >> 1)) * 32 + ((x & 3) << 3 | (y & 3) << 1 | (z & 1))
    assert b[off] == v[x, y, z]
    acc = torch.ones(shape, device="cuda")
    da.unbrick_volume(b, shape, out=acc, accumulate=True)
    assert torch.equal(acc, v + 1)


def test_bricked_volume_parameter(da, vols):
    # a learnable volume kept bricked: gradients come back bricked and match the dense path
    n, S, alpha = 64, 120, 1e-3
    src, dirs = pose_ring(n, 3, 8)
    dense = cuda(vols[n]).requires_grad_(True)
    bv = da.BrickedVolume.from_dense(cuda(vols[n]))
    bv.data.requires_grad_(True)
    s, d = torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda()
    f1 = da.render_poses(dense, s, d, S, alpha, sampler="trilinear", layout="canonical")
    f2 = da.render_poses(bv, s, d, S, alpha, sampler="trilinear")
    assert torch.equal(f1, f2)
    (f1 ** 2).sum().backward()
    (f2 ** 2).sum().backward()
    g2 = da.unbrick_volume(bv.data.grad, bv.shape)
    assert maxnorm_rel(g2.cpu().numpy(), dense.grad.cpu().numpy()) < 1e-5
    assert torch.equal(bv.to_dense(), dense.detach())


def test_brick_cache_tracks_inplace_updates(da, vols):
    n, S = 64, 60
    src, dirs = pose_ring(n, 2, 8)
    v = cuda(vols[n])
    s, d = torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda()
    f1 = da.render_poses(v, s, d, S, 1e-3, layout="bricked")
    v.mul_(1.0).add_(torch.where(v > 1e6, 5e4, 0.0))       # in-place edit bumps the version counter
    f2 = da.render_poses(v, s, d, S, 1e-3, layout="bricked")
    f3 = da.render_poses(v, s, d, S, 1e-3, layout="canonical")
    assert torch.equal(f2, f3) and not torch.equal(f1, f2)


# ----------------------------------------------------------------------------- end to end: pose registration
def test_pose_registration_by_gradient_descent(da):
    # what the reference's `[NW] alignement` notebook attempts and cannot do (no pose gradient):
    # recover a perturbed probe pose by descending d loss / d (apex, median angle) through the HIP backward
    n, R, S, alpha = 64, 48, 96, 1e-3
    u = np.arange(n, dtype=np.float64) / (n - 1)
    smooth = 1.6e6 + 3e5 * np.sin(6 * u)[:, None, None] * np.cos(5 * u)[None, :, None] * np.sin(4 * u + 1)[None, None, :]
    vol = cuda(smooth.astype(np.float32))
    true = da.FanPose((20.0, 30.0, 31.3), (0.8, 0.6), 0.9, R).cuda()
    with torch.no_grad():
        src, dirs = true()
        target = da.render_poses(vol, src, dirs, S, alpha, sampler="trilinear")
    pose = da.FanPose((21.2, 28.9, 31.9), (0.74, 0.67), 0.9, R).cuda()
    opt = torch.optim.Adam(pose.parameters(), lr=0.05)
    losses = []
    for _ in range(150):
        opt.zero_grad()
        src, dirs = pose()
        f = da.render_poses(vol, src, dirs, S, alpha, sampler="trilinear")
        loss = ((f - target) ** 2).sum()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.05 * losses[0], (losses[0], losses[-1])
    assert torch.linalg.norm(pose.apex.detach() - true.apex.detach()) < 0.6


def test_gaussian_pulse_golden(da):
    g = load_golden("g13_gaussian_pulse")
    r = cuda(g["r"])
    for j in range(3):
        length, sigma = (int(v) for v in g[f"p{j}"])
        np.testing.assert_array_equal(da.gaussian_pulse(length, sigma), g[f"pulse{j}"])
        out = da.compute_gaussian_pulse(r, length=length, sigma=sigma)
        assert out.shape == g[f"out{j}"].shape
        assert maxnorm_rel(out.cpu().numpy(), g[f"out{j}"]) < 1e-4


def test_train_impedance_mlp_through_the_renderer(da):
    # miniature of `[DEMO] Train MRI to Impedance MLP - GPU` cell 16: a 1->32->32->1 MLP maps MRI intensity to
    # impedance, frames are rendered through it, and the loss gradient flows HIP backward -> torch autograd -> MLP
    torch.manual_seed(0)
    n, R, S = 32, 24, 48
    mri = torch.from_numpy((phantom(n) / 6.4e6).astype(np.float32)).cuda()
    src, dirs = pose_ring(n, 4, R)
    src, dirs = torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda()

    def make():
        return torch.nn.Sequential(torch.nn.Linear(1, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                                   torch.nn.Linear(32, 1)).cuda()

    def frames(model):
        Z = (model(mri.reshape(-1, 1)).reshape(n, n, n) + 1.5) * 1e6       # impedance volume
        return da.render_poses(Z, src, dirs, S, 1e-3, sampler="trilinear")

    teacher = make()
    with torch.no_grad():
        target = frames(teacher)
    student = make()
    opt = torch.optim.Adam(student.parameters(), lr=3e-3)
    losses = []
    for _ in range(60):
        opt.zero_grad()
        loss = ((frames(student) - target) ** 2).mean()
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in student.parameters())
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])


def test_demo_notebook_call_sequence(da):
    # `[DEMO] REUBEN DATA 46` cells 11-14 with the import swapped (examples/reuben_like_demo.py)
    import importlib.util
    import os as _os
    spec = importlib.util.spec_from_file_location(
        "reuben_like_demo", _os.path.join(_os.path.dirname(_os.path.dirname(__file__)), "examples", "reuben_like_demo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    x, y, z, I, img = mod.run(n=128, n_rays=64, d1=30, d2=120, seed=1)
    assert I.shape == (64, 120 - 18) and I.dtype == torch.float64 and x.dtype == torch.int64
    assert img.shape == (128, 128) and img.dtype == torch.float32
    assert torch.isfinite(img).all() and float(img.abs().max()) > 0
