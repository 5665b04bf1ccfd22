"""The probe-pose parameterisation on the device (csrc/pose.hip: diffus_fan_pose_fwd / _bwd) against the same map composed
from torch ops in float64 (fan_directions_torch + rotation_from_rotvec, themselves pinned by golden G8 / G12 and the CPU
tests of test_host_logic.py): values and the adjoint, single and batched, with and without a rotation vector."""
import math

import numpy as np
import pytest
import torch


def _ref64(median, opening, rotvec, R):
    from diffus_amd.cone import fan_directions_torch, rotation_from_rotvec
    m = median.detach().double().cpu().requires_grad_(True)
    o = opening.detach().double().cpu().requires_grad_(True)
    r = None if rotvec is None else rotvec.detach().double().cpu().requires_grad_(True)
    rows = []
    for p in range(m.shape[0]):
        d = fan_directions_torch(m[p], o[p] if o.dim() else o, R)
        rows.append(d if r is None else d @ rotation_from_rotvec(r[p]).T)
    return torch.stack(rows), m, o, r


@pytest.mark.gpu
@pytest.mark.parametrize("R", [1, 2, 7, 256, 1000])
@pytest.mark.parametrize("case", ["plain", "identity", "small", "tilt", "large", "per_pose_opening"])
def test_fan_pose_kernels_match_the_torch_composition(R, case):
    import diffus_amd as da
    g = torch.Generator().manual_seed(R * 7 + len(case))
    P = 5
    median = (torch.rand(P, generator=g) * 6.0 - 3.0)
    opening = torch.tensor(1.05) if case != "per_pose_opening" else torch.rand(P, generator=g) + 0.3
    rotvec = {"plain": None, "identity": torch.zeros(P, 3), "small": 1e-4 * torch.randn(P, 3, generator=g),
              "tilt": 0.1 * torch.randn(P, 3, generator=g), "large": 2.0 * torch.randn(P, 3, generator=g),
              "per_pose_opening": 0.3 * torch.randn(P, 3, generator=g)}[case]
    ref, m64, o64, r64 = _ref64(median, opening, rotvec, R)
    up = torch.randn(P, R, 3, generator=g, dtype=torch.float64).float().double()   # what the kernel is handed, exactly
    (ref * up).sum().backward()
    m = median.cuda().requires_grad_(True)
    o = opening.cuda().requires_grad_(True)
    r = None if rotvec is None else rotvec.cuda().requires_grad_(True)
    d = da.fan_directions(m, o, R, r)
    assert d.shape == (P, R, 3) and d.dtype == torch.float32 and d.is_cuda
    np.testing.assert_allclose(d.detach().cpu().numpy(), ref.detach().numpy(), atol=1.2e-7, rtol=0)   # one float32 rounding of |.| <= 1
    if rotvec is None:
        assert torch.all(d[..., 2] == 0) and not torch.any(torch.signbit(d[..., 2]))
    (d * up.float().cuda()).sum().backward()

    def close(a, b, what):
        a, b = a.detach().cpu().double().numpy(), b.detach().numpy()
        scale = max(np.abs(b).max(), 0.03 * math.sqrt(R))   # sums of P R terms of size ~1 that cancel, rounded to float32 per pose
        assert np.abs(a - b).max() <= 2e-6 * scale + 1e-6, (what, case, R, np.abs(a - b).max(), scale)

    close(m.grad, m64.grad, "median")
    close(o.grad, o64.grad, "opening")
    if r is not None:
        close(r.grad, r64.grad, "rotvec")


@pytest.mark.gpu
def test_fan_pose_single_pose_and_reference_fan():
    """A scalar median angle gives (R, 3); without a rotation vector the fan is generate_cone_directions((cos m, sin m), ...)
    (reference src/cone.py:242-258, golden G8) to one float32 rounding, third components exact zeros."""
    import diffus_amd as da
    for m, op, R in [(0.3, math.radians(60.0), 256), (-2.0, 0.9, 64), (3.1, 0.2, 3)]:
        d = da.fan_directions(torch.tensor(m).cuda(), op, R)
        want = da.generate_cone_directions(np.array([math.cos(m), math.sin(m)]), op, R)
        assert d.shape == (R, 3)
        np.testing.assert_allclose(d.cpu().numpy(), want.numpy(), atol=1.2e-7, rtol=0)
        assert torch.all(d[:, 2] == 0)
    pose = da.FanPose((20.0, 30.0, 31.0), (0.8, 0.6), 0.9, 48).cuda()
    src, dirs = pose()
    assert dirs.is_cuda and getattr(dirs, "_diffus_planar", False) and dirs.requires_grad
    pose6 = da.FanPose((20.0, 30.0, 31.0), (0.8, 0.6), 0.9, 48, rotvec=(0.05, -0.02, 0.0)).cuda()
    _, d6 = pose6()
    d6.square().sum().backward()
    assert pose6.rotvec.grad is not None and pose6.median_angle.grad is not None
    assert float(pose6.rotvec.grad.abs().max()) < 1e-4          # |dir| = 1 whatever the rotation: no gradient


@pytest.mark.gpu
def test_fan_pose_abi_argument_checks():
    from diffus_amd import _lib
    lib = _lib.load()
    t = torch.zeros(16, device="cuda")
    p = t.data_ptr()
    einval = lib.diffus_fan_pose_fwd(None, p, 0, None, 1, 4, p, None)
    assert einval != 0
    assert lib.diffus_fan_pose_fwd(p, p, 2, None, 1, 4, p, None) == einval          # stride is 0 or 1
    assert lib.diffus_fan_pose_fwd(p, p, 0, None, 0, 4, p, None) == einval
    assert lib.diffus_fan_pose_bwd(p, p, 0, None, p, 1, 4, p, None, p, None) == einval   # a rotvec gradient without a rotvec
    assert lib.diffus_fan_pose_bwd(p, p, 0, None, p, 1, 4, None, None, None, None) == 0   # nothing asked for
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_pose_sweep_registration_batched(da=None):
    """P poses optimised together: FanPose with (P,3) apexes feeds render_poses' pose batch; every pose's loss falls."""
    import diffus_amd as da
    n, R, S, alpha, P = 64, 32, 96, 1e-3, 4
    u = np.arange(n, dtype=np.float64) / (n - 1)
    smooth = 1.6e6 + 3e5 * np.sin(6 * u)[:, None, None] * np.cos(5 * u)[None, :, None] * np.sin(4 * u + 1)[None, None, :]
    vol = torch.from_numpy(smooth.astype(np.float32)).cuda()
    phis = np.linspace(0.2, 1.2, P)
    looks = np.stack([np.cos(phis), np.sin(phis)], 1)
    apex = np.stack([32 - 14 * looks[:, 0], 32 - 14 * looks[:, 1], 31.3 + np.arange(P)], 1)
    true = da.FanPose(apex, looks, 0.9, R, rotvec=np.zeros((P, 3))).cuda()
    with torch.no_grad():
        target = da.render_poses(vol, *true(), S, alpha, sampler="trilinear")
    rng = np.random.default_rng(5)
    pose = da.FanPose(apex + rng.normal(0, 0.8, (P, 3)), looks, 0.9, R, rotvec=rng.normal(0, 0.03, (P, 3))).cuda()
    opt = torch.optim.Adam([{"params": [pose.apex], "lr": 0.05}, {"params": [pose.median_angle, pose.rotvec], "lr": 0.004}])
    first = last = None
    for it in range(200):
        opt.zero_grad()
        f = da.render_poses(vol, *pose(), S, alpha, sampler="trilinear")
        per_pose = ((f - target) ** 2).sum(dim=(1, 2))
        per_pose.sum().backward()
        opt.step()
        if it == 0:
            first = per_pose.detach().cpu()
        last = per_pose.detach().cpu()
    assert torch.all(last < 0.1 * first), (first, last)


def test_batched_fan_pose_on_host_equals_per_pose():
    """Host tensors take the torch composition: a batch is the stack of its poses (and carries gradients)."""
    import diffus_amd as da
    P, R = 3, 9
    looks = np.array([[1.0, 0.0], [0.0, 1.0], [0.6, 0.8]])
    rv = np.array([[0.0, 0.0, 0.0], [0.1, -0.05, 0.02], [0.5, 0.4, -0.3]])
    batch = da.FanPose(np.arange(9.0).reshape(3, 3), looks, 0.9, R, rotvec=rv)
    src, dirs = batch()
    assert src.shape == (P, 3) and dirs.shape == (P, R, 3)
    for p in range(P):
        one = da.FanPose(np.arange(9.0).reshape(3, 3)[p], looks[p], 0.9, R, rotvec=rv[p])
        np.testing.assert_allclose(dirs[p].detach().numpy(), one()[1].detach().numpy(), atol=1e-7)
    dirs[..., 0].sum().backward()
    assert batch.rotvec.grad.shape == (P, 3) and batch.median_angle.grad.shape == (P,)


@pytest.mark.gpu
def test_fan_pose_kernel_against_the_reference_fans_of_g8():
    """Golden G8 = `generate_cone_directions` of the REFERENCE (src/cone.py:242-258) run on seven (direction, opening, n)
    triples: the device kernel, given the median ANGLE of each direction (the angle is rounded to float32 on its way in:
    <= 1.9e-7 rad), reproduces every fan to 3e-7 -- and to one float32 rounding of the float64 evaluation at that angle."""
    import os
    import diffus_amd as da
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_cone_directions.npz"))
    for c in range(int(g["ncases"])):
        d = np.asarray(g[f"c{c}_direction"], dtype=np.float64)
        m = math.atan2(d[1], d[0])
        op, n = float(g[f"c{c}_opening"]), int(g[f"c{c}_n"])
        got = da.fan_directions(torch.tensor(m, dtype=torch.float32).cuda(), op, n).cpu().numpy()
        want = g[f"c{c}_out"]
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, atol=3e-7, rtol=0, err_msg=f"case {c}")
        assert np.all(got[:, 2] == 0)
