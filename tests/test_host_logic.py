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
