"""Host-side logic that needs no GPU: start resolution, fan geometry, phantom, sharding."""
import math

import numpy as np
import pytest
import torch

from diffus_amd import fan_directions_torch, generate_cone_directions, resolve_start
from diffus_amd.distributed import shard_bounds
from diffus_amd.phantom import AIR, BONE, phantom, pose_ring


def test_resolve_start_mirrors_reference():
    # reference src/renderer.py:237-240
    assert resolve_start(0, 256) == 0
    assert resolve_start(-5, 256) == 0
    assert resolve_start(40, 185) == 40
    assert resolve_start(0.25, 48) == 12          # a Python float is a fraction of num_samples
    assert resolve_start(0.999, 10) == 9
    assert resolve_start(np.int64(7), 48) == 7    # numpy ints are used as they are (type() is not int)


def test_cone_directions_contract():
    d = generate_cone_directions((-0.3, -0.95), 0.85, 64)
    assert isinstance(d, torch.Tensor) and d.dtype == torch.float32 and d.shape == (64, 3)
    assert torch.all(d[:, 2] == 0)
    assert torch.allclose(d.norm(dim=1), torch.ones(64), atol=1e-6)
    ang = torch.atan2(d[:, 1], d[:, 0])
    span = (ang[-1] - ang[0]).item() % (2 * math.pi)
    assert abs(span - 0.85) < 1e-5
    # accepts tensors and 3-vectors like the demos do (reference cone.py:249 uses [:2])
    d2 = generate_cone_directions(torch.tensor([-0.3, -0.95, 15.0]), 0.85, 64)   # float32 input: normalised in f32
    assert torch.allclose(d, d2, atol=1e-6)


def test_fan_directions_torch_matches_and_differentiates():
    m = torch.tensor(math.atan2(-0.95, -0.3), dtype=torch.float64, requires_grad=True)
    op = torch.tensor(0.85, dtype=torch.float64, requires_grad=True)
    f = fan_directions_torch(m, op, 64)
    ref = generate_cone_directions((-0.3, -0.95), 0.85, 64).double()
    assert torch.allclose(f, ref, atol=1e-6)
    (f[:, 0].sum() + 2 * f[:, 1].sum()).backward()
    assert m.grad is not None and op.grad is not None and torch.isfinite(m.grad)


def test_phantom_is_deterministic_and_structured():
    a, b = phantom(32), phantom(32)
    assert a.dtype == np.float32 and a.shape == (32, 32, 32) and np.array_equal(a, b)
    assert a[0, 0, 0] == AIR and a.max() > 0.9 * BONE
    assert np.all(a > 0)                            # no negative impedance anywhere
    c = phantom(32, variant=1)
    assert not np.array_equal(a, c)
    s, d = pose_ring(256, 32, 256)
    assert s.shape == (32, 3) and d.shape == (32, 256, 3) and s.dtype == np.float32
    assert np.all(np.abs(np.linalg.norm(d, axis=2) - 1) < 1e-6)
    assert np.all(np.linalg.norm(s[:, :2] - 128, axis=1) < 0.31 * 256)


@pytest.mark.parametrize("P,world", [(256, 8), (32, 1), (10, 4), (3, 8), (0, 2)])
def test_shard_bounds_partition(P, world):
    spans = [shard_bounds(P, g, world) for g in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == P
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b == c and b >= a and d >= c
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1


def test_g12_pose_helpers_match_reference():
    import diffus_amd
    from conftest import load_golden
    g = load_golden("g12_pose_helpers")
    for j in range(int(g["nl"])):
        r = diffus_amd.compute_us_apex_and_direction(*g[f"l{j}_in"])
        np.testing.assert_array_equal(np.array(r["apex"]), g[f"l{j}_apex"])
        assert r["opening_angle"] == g[f"l{j}_opening"]
        np.testing.assert_array_equal(r["direction_vector"], g[f"l{j}_dir"])
    for j in range(int(g["na"])):
        a, d = diffus_amd.cone_us_to_mri_world(g[f"a{j}_apex"], g[f"a{j}_d2"], g[f"a{j}_Au"], g[f"a{j}_At"])
        np.testing.assert_array_equal(a, g[f"a{j}_apex_t1"])
        np.testing.assert_array_equal(d, g[f"a{j}_dir_t1"])
        np.testing.assert_array_equal(diffus_amd.voxel_to_world(g[f"a{j}_apex"], g[f"a{j}_Au"]), g[f"a{j}_v2w"])
        np.testing.assert_array_equal(diffus_amd.world_to_voxel(g[f"a{j}_apex"], g[f"a{j}_At"]), g[f"a{j}_w2v"])
    with pytest.raises(RuntimeError):
        diffus_amd.compute_us_apex_and_direction(1.0, 0.0, 1.0, 5.0)


def test_g21_point_maps_match_reference():
    """mri_to_us_point / us_to_mri_point (reference src/cone.py:21-59; eight notebooks incl. `[DEMO] REUBEN DATA 46` call
    them): slices and index triples equal to the reference's on random volumes and affines, and its range check."""
    import diffus_amd
    from conftest import load_golden
    g = load_golden("g21_point_maps")
    T1, US, At, Au = g["T1"], g["US"], g["At"], g["Au"]
    for j in range(int(g["n"])):
        sl, idx = diffus_amd.mri_to_us_point(*[int(v) for v in g[f"m2u{j}_in"]], T1, At, US, Au)
        np.testing.assert_array_equal(idx, g[f"m2u{j}_idx"]); np.testing.assert_array_equal(sl, g[f"m2u{j}_slice"])
        sl, idx = diffus_amd.us_to_mri_point(*[int(v) for v in g[f"u2m{j}_in"]], US, Au, T1, At)
        np.testing.assert_array_equal(idx, g[f"u2m{j}_idx"]); np.testing.assert_array_equal(sl, g[f"u2m{j}_slice"])
    with pytest.raises(ValueError, match="T1 : indices are out of range"):
        diffus_amd.mri_to_us_point(24, 0, 0, T1, At, US, Au)


def test_fan_pose_six_degrees_of_freedom():
    """FanPose(rotvec=...): directions = R(rotvec) . fan.  A zero rotation vector reproduces the in-plane fan exactly and has
    a gradient; R is a rotation; tilting about the fan's central ray matches pose_ring(roll_deg=...)."""
    import diffus_amd
    from diffus_amd.phantom import pose_ring
    flat = diffus_amd.FanPose((10.0, 20.0, 30.0), (0.6, 0.8), 0.9, 16)
    six = diffus_amd.FanPose((10.0, 20.0, 30.0), (0.6, 0.8), 0.9, 16, rotvec=(0.0, 0.0, 0.0))
    assert torch.equal(flat()[1], six()[1])
    six()[1][:, 2].sum().backward()                       # d (dim-2 components) / d rotvec at the identity: not zero, not NaN
    assert torch.isfinite(six.rotvec.grad).all() and float(six.rotvec.grad.abs().max()) > 0.1
    Rm = diffus_amd.rotation_from_rotvec(torch.tensor([0.3, -0.2, 0.5], dtype=torch.float64))
    assert torch.allclose(Rm @ Rm.T, torch.eye(3, dtype=torch.float64), atol=1e-14) and abs(float(torch.linalg.det(Rm)) - 1) < 1e-14
    # roll by 20 degrees about the central ray = rotation vector 20 deg x (unit central direction)
    n, P, R = 64, 8, 16
    src, dirs = pose_ring(n, P, R, roll_deg=20.0)
    p = 3
    phi = 2 * np.pi * p / P
    look = np.array([-np.cos(phi), -np.sin(phi), 0.0])
    fp = diffus_amd.FanPose(src[p], look[:2], np.radians(60.0), R, rotvec=np.radians(20.0) * look)
    # (pose_ring turns `side` towards +dim 2: a right-handed turn about `look` by +20 degrees does the same)
    assert torch.allclose(fp()[1], torch.from_numpy(dirs[p]), atol=1e-6)


def test_in_plane_fan_pose_carries_the_planar_hint():
    """FanPose without a rotation vector marks its directions as planar (exact zeros in dim 2), so that a pose optimisation of
    an in-plane fan -- directions that require grad, which the renderer never reads back -- still gets the planar scatter launch;
    a 6-DoF pose, or anything computed from the directions afterwards, does not carry the mark."""
    import diffus_amd
    from diffus_amd.renderer import _fans_planar
    flat = diffus_amd.FanPose((10.0, 20.0, 30.0), (0.6, 0.8), 0.9, 16)
    six = diffus_amd.FanPose((10.0, 20.0, 30.0), (0.6, 0.8), 0.9, 16, rotvec=(0.0, 0.1, 0.0))
    d = flat()[1]
    assert d.requires_grad and _fans_planar(d) is True and bool((d[:, 2] == 0).all())
    assert _fans_planar(six()[1]) is False                      # a host tensor: looked at directly
    assert not getattr(d * 2.0, "_diffus_planar", False)


def test_fan_pose_module():
    import diffus_amd
    fp = diffus_amd.FanPose((88.0, -11.5, 110.0), (-0.3, -0.95), 0.85, 64, learn_opening=True)
    src, dirs = fp()
    ref = generate_cone_directions((-0.3, -0.95), 0.85, 64)
    assert torch.allclose(dirs, ref, atol=2e-6) and src.shape == (3,)
    (dirs[:, 0].sum() + src.sum()).backward()
    assert fp.median_angle.grad is not None and fp.opening_angle.grad is not None and fp.apex.grad is not None


def test_layout_fits_follows_the_abi_limits():
    from diffus_amd.renderer import layout_fits
    assert all(layout_fits(k, (256, 256, 256)) for k in ("canonical", "bricked", "paired"))
    assert all(layout_fits(k, (512, 512, 512)) for k in ("canonical", "bricked", "paired"))
    assert layout_fits("paired", (64, 640, 640)) and not layout_fits("paired", (8, 1024, 1024))      # brick-row stride >= 2^24 bytes
    assert layout_fits("bricked", (8, 1024, 512)) and not layout_fits("bricked", (8, 2048, 1024))
    assert layout_fits("canonical", (8, 2048, 1024))
    assert not layout_fits("canonical", (1024, 1024, 1024))                                       # 2^30 floats: 32-bit offsets


def test_host_pose_cache_sees_numpy_side_edits():
    """The upload cache of host poses is keyed by identity + version + CONTENT: a tensor made by torch.from_numpy can be
    rewritten through its array without its version counter moving."""
    from diffus_amd import renderer as R
    arr = np.linspace(0, 1, 12, dtype=np.float32).reshape(4, 3)
    dirs = torch.from_numpy(arr)
    src = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
    dev = torch.device("cpu")
    R._host_pose_store(dev, src, dirs, "dsrc", "ddirs", True)
    assert R._host_pose_lookup(dev, src, dirs) == ("dsrc", "ddirs", True)
    v = dirs._version
    arr[2, 1] += 0.5                                   # NumPy-side edit: same tensor object, same version
    assert dirs._version == v
    assert R._host_pose_lookup(dev, src, dirs) is None
    R._host_pose_store(dev, src, dirs, "dsrc2", "ddirs2", False)
    assert R._host_pose_lookup(dev, src, dirs) == ("dsrc2", "ddirs2", False)
    dirs.add_(1.0)                                     # torch-side edit: the version moves
    assert R._host_pose_lookup(dev, src, dirs) is None
    assert R._fingerprint(torch.empty(0)) == R._fingerprint(torch.empty(0))
