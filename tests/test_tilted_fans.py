"""Fans that are NOT planar in dim 2, at full size.

`plot_beam_frame` takes any `directions` (src/renderer.py:201-217, :119-124) and a probe-pose optimisation
(`notebooks/[NW] alignement.ipynb` cells 13-14) produces exactly such fans: the plane tilted out of the slice (roll about
the central ray), the central ray lifted out of it (pitch), or the fan lying in another coordinate plane altogether.
Config-2 shape (256 rays x 512 steps, 256^3): frame <= 2e-5 against the float64 restatement at the float32 sample points
of the reference, d/dvolume, d/dsource, d/ddirections <= 1e-3; and the per-voxel accumulation bound of the
scatter on these patches (same formula as test_fixed_point_scatter_error_bound_per_voxel).
"""
import numpy as np
import pytest
import torch

from conftest import maxnorm_rel
from diffus_amd.phantom import phantom, pose_ring

pytestmark = pytest.mark.gpu

TILTS = [  # roll, pitch, plane, pose of the 32-ring
    (5.0, 0.0, (0, 1), 3),
    (20.0, 0.0, (0, 1), 3),
    (45.0, 0.0, (0, 1), 11),
    (0.0, 20.0, (0, 1), 7),
    (20.0, 10.0, (0, 1), 18),
    (0.0, 0.0, (0, 2), 3),
    (0.0, 0.0, (1, 2), 5),
    (20.0, 0.0, (0, 2), 3),
]


@pytest.fixture(scope="module")
def da():
    import diffus_amd
    from diffus_amd import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return diffus_amd


@pytest.fixture(scope="module")
def vol256():
    return phantom(256)


@pytest.mark.parametrize("roll,pitch,plane,pose", TILTS)
def test_tilted_fan_full_size_values_vs_float64_autograd(da, vol256, roll, pitch, plane, pose):
    from oracle import autograd_ref as ar
    n, R, S, alpha = 256, 256, 512, 1e-4
    src, dirs = pose_ring(n, 32, R, roll_deg=roll, pitch_deg=pitch, plane=plane)
    vol = torch.from_numpy(vol256).cuda().requires_grad_(True)
    s = torch.from_numpy(src[pose:pose + 1]).cuda().requires_grad_(True)
    d = torch.from_numpy(dirs[pose:pose + 1]).cuda().requires_grad_(True)
    f = da.render_poses(vol, s, d, S, alpha, sampler="trilinear")
    (f ** 2).sum().backward()
    got = (f.detach()[0].cpu().numpy(), s.grad[0].cpu().numpy(), d.grad[0].cpu().numpy())
    gv = vol.grad.cpu()
    del vol, f
    v64 = torch.from_numpy(vol256).double().requires_grad_(True)
    s64 = torch.from_numpy(src[pose]).double().requires_grad_(True)
    d64 = torch.from_numpy(dirs[pose]).double().requires_grad_(True)
    fr = ar.render(v64, s64, d64, S, alpha, 0, "trilinear", points="f32")
    (fr ** 2).sum().backward()
    e_f = maxnorm_rel(got[0], fr.detach().numpy())
    e_s = maxnorm_rel(got[1], s64.grad.numpy())
    e_d = maxnorm_rel(got[2], d64.grad.numpy())
    gref = v64.grad
    den = float(gref.abs().max())
    e_v = max(float((gv[i0:i0 + 64].double() - gref[i0:i0 + 64]).abs().max()) for i0 in range(0, n, 64)) / den
    print("tilt roll %.0f pitch %.0f plane %s: frame %.2e gsrc %.2e gdir %.2e gvol %.2e" % (roll, pitch, plane, e_f, e_s, e_d, e_v))
    assert e_f < 2e-5
    assert e_s < 1e-3 and e_d < 1e-3
    assert e_v < 1e-3
    assert int((gv != 0).sum()) > 0.5 * int((gref != 0).sum())


@pytest.mark.parametrize("sampler", ["trilinear", "nearest"])
@pytest.mark.parametrize("alpha", [1e-4, 0.5])
@pytest.mark.parametrize("roll,pitch,plane", [(20.0, 0.0, (0, 1)), (45.0, 10.0, (0, 1)), (0.0, 0.0, (0, 2)), (5.0, 0.0, (1, 2))])
def test_scatter_error_bound_per_voxel_oblique(da, alpha, sampler, roll, pitch, plane):
    """test_fixed_point_scatter_error_bound_per_voxel on fans that leave the slice: the kernel's own zbar scattered
    exactly (float64, CPU) against what the scatter kernel left in the gradient, per voxel relative to the voxel's mass
    m_v = sum_s |zbar_s| w_sv:  |g_v - exact_v| <= 4e-6 m_v + 2^-40 max m  -- with the reference's default
    attenuation_coeff = 0.5 (zbar falls by 2^-46 across a patch) as well as 1e-4."""
    from diffus_amd import CapturedStep, _lib
    from oracle import autograd_ref as ar
    n, R, S = 64, 64, 96
    v = phantom(n)
    src, dirs = pose_ring(n, 8, R, roll_deg=roll, pitch_deg=pitch, plane=plane)
    hp = CapturedStep(torch.from_numpy(v).cuda(), torch.from_numpy(src[2:3]).cuda(), torch.from_numpy(dirs[2:3]).cuda(), S, alpha,
                      sampler, persistent=False)
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
    print("oblique per-voxel bound: roll %.0f pitch %.0f plane %s alpha %g %s: worst %.3g of the bound" % (roll, pitch, plane, alpha, sampler, worst))
    assert worst <= 1.0, worst
    deep = (mass < 2.0 ** -30 * float(mass.max())) & (mass > 2.0 ** -38 * float(mass.max()))
    if alpha == 0.5:
        assert int(deep.sum()) > 10
        assert int((g[deep] != 0).sum()) > 0.9 * int(deep.sum())


def _ring_step(Step, vol, src, dirs, S, **kw):
    st = Step(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), S, 1e-3, "trilinear", **kw)
    st.step()
    torch.cuda.synchronize()
    return st


def test_fans_hint_never_changes_the_result(da):
    """DIFFUS_FANS_PLANAR is a hint: a wrong promise (tilted fans on the planar launch: the general 3-D tile takes them) and
    no promise at all (planar fans on the slab-capable launch) give the gradients of the right launch; `fans="auto"` looks at
    the directions once, and set_poses() with new directions makes the answer unknown (the slab-capable launch)."""
    from diffus_amd import CapturedStep
    n, P, R, S = 64, 4, 64, 96
    vol = torch.from_numpy(phantom(n)).cuda()
    flat = pose_ring(n, P, R)
    tilt = pose_ring(n, P, R, roll_deg=20.0, pitch_deg=5.0)
    for (src, dirs), planar in ((flat, True), (tilt, False)):
        a = _ring_step(CapturedStep, vol, src, dirs, S, fans="planar")
        b = _ring_step(CapturedStep, vol, src, dirs, S, fans="oblique")
        c = _ring_step(CapturedStep, vol, src, dirs, S)
        assert a.fans_planar and not b.fans_planar and c.fans_planar == planar
        den = float(b.gvol.abs().max())
        assert den > 0
        assert float((a.gvol - b.gvol).abs().max()) <= 2e-5 * den          # (the 3-D tile is 32-bit fixed point: 2^-20 of a patch's largest)
        assert float((c.gvol - b.gvol).abs().max()) <= 2e-5 * den
        assert torch.equal(a.frame, b.frame) and torch.equal(a.gsrc, b.gsrc) and torch.equal(a.gdirs, b.gdirs)
    st = _ring_step(CapturedStep, vol, flat[0], flat[1], S)
    assert st.fans_planar
    st.set_poses(torch.from_numpy(tilt[0]).cuda(), torch.from_numpy(tilt[1]).cuda())
    assert not st.fans_planar
    st.step()
    torch.cuda.synchronize()
    ref = _ring_step(CapturedStep, vol, tilt[0], tilt[1], S, fans="oblique")
    assert float((st.gvol - ref.gvol).abs().max()) <= 1e-6 * float(ref.gvol.abs().max())


@pytest.mark.parametrize("sampler", ["trilinear", "nearest"])
@pytest.mark.parametrize("step", [1.0, 1.7, 2.6, 6.0, 19.0])
def test_slab_scatter_long_steps_every_chunking(da, sampler, step):
    """The slab path keeps a pass in one tile when its column box x layers fits, walks it in chunks of rows when it does not,
    and adds straight to memory when a single row is too wide.  Step lengths 1.7 ... 19 voxels on a tilted fan in a wide
    volume walk through all of these; every one against float64 autograd of the same render (the reference accepts any
    direction norm: renderer.py:94-110)."""
    from oracle import autograd_ref as ar
    rng = np.random.default_rng(13)
    v = (1.5e6 + 2e5 * rng.standard_normal((300, 300, 40))).astype(np.float32)
    R, S = 64, 96
    ang = np.linspace(0.35, 1.2, R)
    tilt = 0.3
    dirs = (step * np.stack([np.cos(ang), np.sin(ang) * np.cos(tilt), np.sin(ang) * np.sin(tilt)], 1)).astype(np.float32)
    src = np.array([6.3, 9.1, 3.4], np.float32)
    v64 = torch.from_numpy(v).double().requires_grad_(True)
    f64 = ar.render(v64, torch.from_numpy(src).double(), torch.from_numpy(dirs).double(), S, 2e-3, 0, sampler, points="f32")
    up = torch.randn(f64.shape, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    (f64 * up).sum().backward()
    gv_ref = v64.grad.numpy()
    for layout in ("paired", "canonical"):
        vol = torch.from_numpy(v).cuda().requires_grad_(True)
        f = da.render_poses(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), S, 2e-3, sampler=sampler, layout=layout)[0]
        assert maxnorm_rel(f.detach().cpu().numpy(), f64.detach().numpy()) < 2e-5
        (f * up.float().cuda()).sum().backward()
        assert maxnorm_rel(vol.grad.cpu().numpy(), gv_ref) < 1e-3, (layout, step)
        assert abs(float(vol.grad.double().sum()) - gv_ref.sum()) <= 1e-4 * np.abs(gv_ref).sum()


def test_six_dof_pose_registration_by_gradient_descent(da):
    """What `notebooks/[NW] alignement.ipynb` cells 13-14 attempt (and the reference cannot do: no pose gradient),
    with all six degrees of freedom: a probe pose perturbed by a 5 degree tilt out of the slice (roll about the central ray
    + pitch) and 3 voxels of apex offset is recovered by descending d loss / d (apex, median angle, rotation vector) through
    the HIP backward -- d/d directions in 3-D, carried to the rotation vector by FanPose (which generalises
    src/cone.py:187-209, :242-258: z = 0 there)."""
    n, R, S, alpha = 64, 48, 96, 1e-3
    u = np.arange(n, dtype=np.float64) / (n - 1)
    smooth = 1.6e6 + 3e5 * np.sin(6 * u)[:, None, None] * np.cos(5 * u)[None, :, None] * np.sin(4 * u + 1)[None, None, :]
    vol = torch.from_numpy(smooth.astype(np.float32)).cuda()
    true = da.FanPose((20.0, 30.0, 31.3), (0.8, 0.6), 0.9, R, rotvec=(0.0, 0.0, 0.0)).cuda()
    with torch.no_grad():
        src, dirs = true()
        target = da.render_poses(vol, src, dirs, S, alpha, sampler="trilinear")
    look = np.array([0.8, 0.6, 0.0])
    side = np.array([-0.6, 0.8, 0.0])
    tilt = np.radians(4.0) * look + np.radians(3.0) * side             # 5 degrees in all, out of the slice both ways
    pose = da.FanPose((21.8, 28.1, 32.8), (0.8, 0.6), 0.9, R, rotvec=tilt).cuda()   # |apex offset| = 3.0 voxels
    assert abs(float(torch.linalg.norm(pose.apex.detach() - true.apex.detach())) - 3.0) < 0.1
    opt = torch.optim.Adam([{"params": [pose.apex], "lr": 0.05}, {"params": [pose.median_angle, pose.rotvec], "lr": 0.004}])
    losses = []
    for _ in range(300):
        opt.zero_grad()
        src, dirs = pose()
        f = da.render_poses(vol, src, dirs, S, alpha, sampler="trilinear")
        loss = ((f - target) ** 2).sum()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    Rm = da.rotation_from_rotvec(pose.rotvec.detach().double().cpu())
    # what is left of the rotation, median angle included: the angle between the recovered and the true central rays / planes
    with torch.no_grad():
        d_rec, d_true = pose()[1].double().cpu(), true()[1].double().cpu()
    ang = torch.rad2deg(torch.acos(((d_rec * d_true).sum(1) / (d_rec.norm(dim=1) * d_true.norm(dim=1))).clamp(-1, 1))).max()
    print("6-DoF registration: loss %.3g -> %.3g, apex error %.3f voxels, worst ray angle %.3f deg" % (
        losses[0], losses[-1], float(torch.linalg.norm(pose.apex.detach() - true.apex.detach())), float(ang)), Rm.shape)
    assert losses[-1] < 0.05 * losses[0], (losses[0], losses[-1])
    assert torch.linalg.norm(pose.apex.detach() - true.apex.detach()) < 0.6
    assert float(ang) < 1.0


@pytest.mark.parametrize("sampler", ["trilinear", "nearest"])
@pytest.mark.parametrize("start,f64", [(7, False), (0, True), (12, True)])
def test_tilted_fan_start_crop_and_f64_poses(da, sampler, start, f64):
    """The slab path with a start crop (the cropped rays' steps begin at `start`; the per-pose median replaces the first kept
    coefficient, reference :237-244) and with float64 poses (torch's promotion of source + k * direction, :119-124): frame and
    gradients of a 64-ray tilted fan against float64 autograd of the restatement."""
    from oracle import autograd_ref as ar
    n, R, S, alpha = 48, 64, 120, 2e-3
    rng = np.random.default_rng(5)
    v = (1.5e6 + 2e5 * rng.standard_normal((n, n, n))).astype(np.float32)
    src, dirs = pose_ring(n, 8, R, roll_deg=25.0, pitch_deg=8.0)
    dt = np.float64 if f64 else np.float32
    s_np, d_np = src[3].astype(dt) + dt(0.123456789), dirs[3].astype(dt)
    v64 = torch.from_numpy(v).double().requires_grad_(True)
    s64 = torch.from_numpy(s_np).double().requires_grad_(True)
    d64 = torch.from_numpy(d_np).double().requires_grad_(True)
    f_ref = ar.render(v64, s64, d64, S, alpha, start, sampler, points="exact" if f64 else "f32")
    up = torch.randn(f_ref.shape, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    (f_ref * up).sum().backward()
    for layout in ("paired", "canonical"):
        vol = torch.from_numpy(v).cuda().requires_grad_(True)
        s = torch.from_numpy(s_np).cuda().requires_grad_(True)
        d = torch.from_numpy(d_np).cuda().requires_grad_(True)
        f = da.render_poses(vol, s, d, S, alpha, start=start, sampler=sampler, layout=layout)[0]
        assert maxnorm_rel(f.detach().cpu().numpy(), f_ref.detach().numpy()) < 5e-5, (layout, start)
        (f * up.to(f.dtype).cuda()).sum().backward()
        assert maxnorm_rel(vol.grad.cpu().numpy(), v64.grad.numpy()) < 1e-3, (layout, start)
        if sampler == "trilinear":
            assert maxnorm_rel(s.grad.cpu().numpy(), s64.grad.numpy()) < 1e-3
            assert maxnorm_rel(d.grad.cpu().numpy(), d64.grad.numpy()) < 1e-3


def test_tilted_fan_config5_shape_values_vs_float64_autograd(da):
    """The slab path at the BASELINE config-5 shape (512 rays x 1024 steps through a 512^3 volume: two waves per ray in the scan,
    33 step groups of patches in the scatter): one fan rolled by 20 degrees and pitched by 5, frame and the three gradients
    against float64 autograd at the float32 sample points."""
    from oracle import autograd_ref as ar
    n, R, S, alpha, pose = 512, 512, 1024, 1e-4, 5
    v = phantom(n, variant=1)
    src, dirs = pose_ring(n, 8, R, roll_deg=20.0, pitch_deg=5.0)
    vol = torch.from_numpy(v).cuda().requires_grad_(True)
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
    fr = ar.render(v64, s64, d64, S, alpha, 0, "trilinear", points="f32")
    (fr ** 2).sum().backward()
    assert maxnorm_rel(got[0], fr.detach().numpy()) < 2e-5
    assert maxnorm_rel(got[1], s64.grad.numpy()) < 1e-3
    assert maxnorm_rel(got[2], d64.grad.numpy()) < 1e-3
    gref = v64.grad
    den = float(gref.abs().max())
    err = max(float((gv[i0:i0 + 64].double() - gref[i0:i0 + 64]).abs().max()) for i0 in range(0, n, 64))
    assert den > 0 and err / den < 1e-3, err / den
    assert int((gv != 0).sum()) > 0.5 * int((gref != 0).sum())


def test_full_size_registration_example_descends(da):
    """examples/register_probe_pose.py at BASELINE config 2's frame (256 rays x 512 steps, 256^3): the loss falls and the pose
    moves towards the true one (all six degrees of freedom through the slab scatter and the HIP pose gradient)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "register_probe_pose", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "register_probe_pose.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    history, apex_err, ang = mod.run(iters=150, report=149)
    assert history[-1][1] < 0.2 * history[0][1], history
    assert apex_err < 2.8 and ang < 2.5, (apex_err, ang)          # from 3.02 voxels and 4.6 degrees; 400 iterations: 1.1 and 0.36
    # the same loop as ONE captured graph per iteration: every launch of it is capturable, and the descent is the same
    hist_g, apex_g, ang_g = mod.run(iters=150, report=149, graph=True)
    assert abs(hist_g[-1][1] - history[-1][1]) < 0.05 * history[-1][1] and abs(apex_g - apex_err) < 0.05, (hist_g, history)
    # ... and with render + loss + backward as the one-pass step (CapturedStep.mse_loss): the same numbers again
    hist_o, apex_o, ang_o = mod.run(iters=150, report=149, graph=True, one_pass=True)
    assert abs(hist_o[-1][1] - hist_g[-1][1]) < 0.02 * hist_g[-1][1] and abs(apex_o - apex_g) < 0.02, (hist_o, hist_g)
    # a sweep of four frames registered together (one FanPose module, one render launch per iteration for all)
    hist_s, apex_s, ang_s = mod.run(iters=120, report=119, graph=True, one_pass=True, poses=4)
    assert hist_s[-1][1] < 0.1 * 4 * history[0][1] and ang_s < 4.0 and apex_s < 2.9, (hist_s, apex_s, ang_s)   # (every frame starts like the single one)


def _coplanar_case(seed, planar=False, where=None):
    """A fan that is a RIGID MOTION of a planar one (every patch coplanar: the slab path), at random: odd volume shapes,
    any orientation (uniform over SO(3); every third seed close to a coordinate plane, where the minor axis flips between
    patches), source inside / on the border / outside, short and long steps, crops, f32 / f64 poses."""
    rng = np.random.default_rng(7000 + seed)
    dims = tuple(int(v) for v in rng.integers(5, 72, size=3))
    if seed % 5 == 1:
        dims = (int(rng.integers(2, 6)), int(rng.integers(30, 90)), int(rng.integers(5, 40)))
    vol = (1.5e6 + 2e5 * rng.standard_normal(dims)).astype(np.float32)
    R = int(rng.choice([1, 2, 5, 31, 32, 33, 40, 64, 70]))
    S = int(rng.choice([3, 17, 33, 64, 65, 96, 130, 200]))
    start = int(rng.integers(1, max(2, S // 3))) if seed % 4 == 3 and S > 4 else 0
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    if seed % 3 == 0:      # nearly axis-aligned planes: a coordinate permutation times a small rotation
        perm = np.eye(3)[list(rng.permutation(3))]
        e = 0.02 * rng.standard_normal(3)
        K = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
        Rm = perm @ (np.eye(3) + K + 0.5 * K @ K)
    if planar:             # (tools/fuzz_slab.py: the same cases with the fan left in the slice, as a control)
        Rm = np.eye(3)
    opening = np.radians(rng.uniform(5, 140))
    ang = rng.uniform(0, 2 * np.pi) + np.linspace(-opening / 2, opening / 2, R)
    fan = np.stack([np.cos(ang), np.sin(ang), np.zeros(R)], 1) @ Rm.T
    step = float(rng.choice([0.15, 0.5, 1.0, 1.0, 1.7, 3.0]))
    dirs = fan * step
    centre = np.array(dims) / 2.0
    where = seed % 4 if where is None else where      # (tools/fuzz_slab.py moves the cropped cases off the volume's corner)
    src = centre + rng.normal(0, 0.25, 3) * np.array(dims) if where < 2 else (
        centre + rng.normal(0, 1.2, 3) * np.array(dims) if where == 2 else np.array([0.3, rng.uniform(0, dims[1] - 1), dims[2] - 1.4]))   # beside two faces (exactly ON one, the d/dpoint convention
    # at p == 0 decides the answer: grid_sample's zero there is pinned by G9, float64 autograd through clamp says one)
    f64 = seed % 6 == 4
    return vol, src.astype(np.float64 if f64 else np.float32), dirs.astype(np.float32), S, start, float(10 ** rng.uniform(-4, -2))


@pytest.mark.parametrize("seed", range(60))
def test_random_coplanar_fans_vs_float64_autograd(da, seed):
    """Seeded sweep of the slab path (and of its hand-over to the general path where a patch is too thick): d/dvolume in every
    gradient layout, d/dsource, d/ddirections against float64 autograd at the float32 sample points, <= 1e-3."""
    from oracle import autograd_ref as ar
    vol, src, dirs, S, start, alpha = _coplanar_case(seed)
    vol = np.abs(vol) + 1e5                             # well-conditioned reflection coefficients (the float32-vs-float64 question is not this test's)
    for sampler in ("trilinear", "nearest"):
        v64 = torch.from_numpy(vol).double().requires_grad_(True)
        s64 = torch.from_numpy(src).double().requires_grad_(True)
        d64 = torch.from_numpy(dirs).double().requires_grad_(True)
        # (float32 poses: the reference's float32 march decides where the samples are -- one ulp elsewhere can put a sample on the
        # other side of a cell boundary, where the trilinear gradient jumps, or flip a nearest index: tools/fuzz_slab.py)
        f64 = ar.render(v64, s64, d64, S, alpha, start, sampler, points="f32" if src.dtype == np.float32 else "exact")
        up = torch.randn(f64.shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
        (f64 * up).sum().backward()
        ref = v64.grad.numpy()
        for layout in ("bricked", "paired", "canonical"):
            v = torch.from_numpy(vol).cuda().requires_grad_(True)
            s = torch.from_numpy(src).cuda().requires_grad_(True)
            d = torch.from_numpy(dirs).cuda().requires_grad_(True)
            f = da.render_poses(v, s, d, S, alpha, start=start, sampler=sampler, layout=layout)[0]
            (f * up.float().cuda()).sum().backward()
            gv = v.grad.cpu().numpy()
            assert np.all(np.isfinite(gv))
            if np.max(np.abs(ref)) < 1e-14:
                assert np.max(np.abs(gv)) < 1e-9, (seed, sampler, layout)
                continue
            assert maxnorm_rel(gv, ref) < 1e-3, (seed, sampler, layout, maxnorm_rel(gv, ref))
            if sampler == "trilinear":
                assert maxnorm_rel(s.grad.cpu().numpy(), s64.grad.numpy()) < 1e-3, (seed, layout)
                assert maxnorm_rel(d.grad.cpu().numpy(), d64.grad.numpy()) < 1e-3, (seed, layout)


@pytest.mark.parametrize("seed", list(range(8)) + list(range(100, 108)))
def test_random_coplanar_fans_medium_size(da, seed):
    """The same sweep at sizes where patches fill their tiles, need several passes (faces + inside) and row chunks: volumes of
    100-200 voxels a side, 96 rays x 400 steps of 1-2.5 voxels, any orientation; trilinear, bricked gradient."""
    from oracle import autograd_ref as ar
    rng = np.random.default_rng(9100 + seed)
    dims = tuple(int(v) for v in rng.integers(100, 201, size=3))
    u = [np.arange(d, dtype=np.float64) / (d - 1) for d in dims]
    vol = (1.5e6 + 2e5 * np.sin(9 * u[0])[:, None, None] * np.cos(7 * u[1])[None, :, None] * np.sin(5 * u[2] + 1)[None, None, :]
           + 3e4 * rng.standard_normal(dims)).astype(np.float32)
    R, S, alpha = 96, 400, 1e-4
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    opening = np.radians(rng.uniform(30, 100))
    ang = rng.uniform(0, 2 * np.pi) + np.linspace(-opening / 2, opening / 2, R)
    dirs = (np.stack([np.cos(ang), np.sin(ang), np.zeros(R)], 1) @ Rm.T * float(rng.choice([1.0, 1.0, 2.5]))).astype(np.float32)
    src = (np.array(dims) / 2.0 + rng.normal(0, 0.2, 3) * np.array(dims)).astype(np.float32)
    v64 = torch.from_numpy(vol).double().requires_grad_(True)
    s64 = torch.from_numpy(src).double().requires_grad_(True)
    d64 = torch.from_numpy(dirs).double().requires_grad_(True)
    f64 = ar.render(v64, s64, d64, S, alpha, 0, "trilinear", points="f32")       # at the float32 march's sample points (see above)
    (f64 ** 2).sum().backward()
    v = torch.from_numpy(vol).cuda().requires_grad_(True)
    s = torch.from_numpy(src).cuda().requires_grad_(True)
    d = torch.from_numpy(dirs).cuda().requires_grad_(True)
    f = da.render_poses(v, s, d, S, alpha, sampler="trilinear", layout="bricked")[0]
    (f ** 2).sum().backward()
    fo = f64.detach().numpy()
    assert maxnorm_rel(f.detach().cpu().numpy(), fo) < 1e-4, seed
    assert maxnorm_rel(v.grad.cpu().numpy(), v64.grad.numpy()) < 1e-3, seed
    assert maxnorm_rel(s.grad.cpu().numpy(), s64.grad.numpy()) < 1e-3, seed
    assert maxnorm_rel(d.grad.cpu().numpy(), d64.grad.numpy()) < 1e-3, seed


@pytest.mark.parametrize("sampler", ["trilinear", "nearest"])
def test_mixed_batch_is_the_sum_of_its_poses(da, sampler):
    """One launch with every kind of fan side by side -- in the slice, rolled, pitched, lying in another coordinate plane, and
    one whose rays are NOT coplanar (the general 3-D tile) -- gives, pose by pose, the frames and pose gradients of the single-pose
    launches, and d/dvolume is the sum of theirs (each block picks its path from its own rays: csrc/scatter.hip)."""
    n, R, S, alpha = 96, 64, 160, 1e-3
    vol_np = phantom(n)
    kinds = [dict(), dict(roll_deg=20.0), dict(pitch_deg=15.0), dict(plane=(0, 2)), dict(roll_deg=45.0, pitch_deg=5.0), dict(plane=(1, 2))]
    srcs, dirss = [], []
    for i, kw in enumerate(kinds):
        s_, d_ = pose_ring(n, 8, R, **kw)
        srcs.append(s_[i]); dirss.append(d_[i].copy())
    rng = np.random.default_rng(3)
    warped = dirss[1].copy()
    warped += 0.05 * rng.standard_normal(warped.shape).astype(np.float32)     # no plane holds these rays
    srcs.append(srcs[1]); dirss.append(warped)
    src, dirs = np.stack(srcs), np.stack(dirss)
    P = len(srcs)
    up = torch.randn(P, R, S, generator=torch.Generator().manual_seed(11)).cuda()

    def run(sel):
        v = torch.from_numpy(vol_np).cuda().requires_grad_(True)
        s = torch.from_numpy(src[sel]).cuda().requires_grad_(True)
        d = torch.from_numpy(dirs[sel]).cuda().requires_grad_(True)
        f = da.render_poses(v, s, d, S, alpha, sampler=sampler, layout="paired")
        (f * up[sel]).sum().backward()
        return f.detach(), v.grad, s.grad, d.grad

    f_all, gv_all, gs_all, gd_all = run(list(range(P)))
    gv_sum = torch.zeros_like(gv_all, dtype=torch.float64)
    for p in range(P):
        f1, gv1, gs1, gd1 = run([p])
        assert torch.equal(f1[0], f_all[p]), p
        if sampler == "trilinear":
            assert torch.equal(gs1[0], gs_all[p]) and torch.equal(gd1[0], gd_all[p]), p
        gv_sum += gv1.double()
    den = float(gv_sum.abs().max())
    assert den > 0
    assert float((gv_all.double() - gv_sum).abs().max()) <= 3e-6 * den       # float32 accumulation order, and the 3-D tile's fixed point
