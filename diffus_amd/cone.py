"""Fan geometry: mirror of the reference's src/cone.py:242-258.

`generate_cone_directions` is bit-exact with the reference (tests/golden G8).
`fan_directions_torch` is its differentiable torch twin, so that the pose
gradients produced by the HIP backward reach (median angle, opening angle).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .phantom import cone_directions_np


def generate_cone_directions(direction_mri_world, opening_angle, n_rays):
    """n_rays unit vectors (n_rays, 3) float32 that fan out symmetrically about the first two components of
    `direction_mri_world`, `opening_angle` radians from the first ray to the last, all with a zero third component --
    signature and values of reference src/cone.py:242-258 (bit-exact, golden G8)."""
    if isinstance(direction_mri_world, torch.Tensor):
        direction_mri_world = direction_mri_world.detach().cpu().numpy()
    return torch.from_numpy(cone_directions_np(direction_mri_world, float(opening_angle), int(n_rays)))


def fan_directions_torch(median_angle: torch.Tensor, opening_angle, n_rays: int) -> torch.Tensor:
    """Differentiable fan: ray a has in-plane angle median_angle + a, with
    a in linspace(-opening/2, opening/2, n_rays).  Equal (to rounding) to
    generate_cone_directions((cos m, sin m), opening, n_rays)."""
    median_angle = torch.as_tensor(median_angle)
    opening_angle = torch.as_tensor(opening_angle, dtype=median_angle.dtype, device=median_angle.device)
    lin = torch.linspace(-0.5, 0.5, n_rays, dtype=median_angle.dtype, device=median_angle.device)
    ang = median_angle + opening_angle * lin
    return torch.stack([torch.cos(ang), torch.sin(ang), torch.zeros_like(ang)], dim=-1)


def median_angle_of(direction) -> float:
    return math.atan2(float(direction[1]), float(direction[0]))


# ---- probe-pose helpers (host-side NumPy geometry) -------------------------------------------
# Values are bit-identical with the reference's helpers (golden G12): the same NumPy primitives on the same operands
# (np.linalg.inv, np.linalg.norm of each 2-vector, np.dot), arranged this package's way.
def _homogeneous(v):
    return np.append(np.asarray(v, dtype=np.float64), 1.0)


def _unit(v):
    return v / np.linalg.norm(v)


def voxel_to_world(idx_ijk, affine):
    """Voxel index (i, j, k) -> world coordinates through a NIfTI affine (what reference src/cone.py:10-13 returns)."""
    return (affine @ _homogeneous(idx_ijk))[:3]


def world_to_voxel(xyz, affine):
    """World coordinates -> (fractional) voxel index: the inverse affine applied (reference src/cone.py:15-19)."""
    return (np.linalg.inv(affine) @ _homogeneous(xyz))[:3]


def compute_us_apex_and_direction(m_left, b_left, m_right, b_right):
    """The fan between the two edge lines y = m x + b picked on the ultrasound slice: where they cross (the apex), the
    angle between them and the unit bisector pointing into the fan.  Same dict as reference src/cone.py:98-126."""
    if np.isclose(m_left, m_right):
        raise RuntimeError("The slopes are nearly equal; no defined intersection.")
    apex_x = (b_right - b_left) / (m_left - m_right)
    apex = (apex_x, m_left * apex_x + b_left)
    # unit vectors along the two edges, away from the apex into the fan (left edge: towards -x, right edge: towards +x)
    edge_l, edge_r = _unit(np.array([-1, -m_left])), _unit(np.array([1, m_right]))
    spread = np.arccos(np.clip(np.dot(edge_l, edge_r), -1.0, 1.0))
    return {"apex": apex, "opening_angle": spread, "direction_vector": _unit(edge_l + edge_r)}


def cone_us_to_mri_world(apex_us_vox, direction_vec_us_2d, US_affine, T1_affine):
    """Carry a fan from ultrasound-voxel to MRI-voxel coordinates: the apex goes through world space (both affines),
    the in-plane direction through the linear parts only, then back to a unit 2-vector.  Same pair as reference
    src/cone.py:187-209: (apex in MRI voxels (3,), unit direction (2,))."""
    apex_mri = world_to_voxel(voxel_to_world(apex_us_vox, US_affine), T1_affine)
    lin_us, lin_mri = US_affine[:3, :3], T1_affine[:3, :3]
    heading = lin_mri @ (np.linalg.inv(lin_us) @ np.append(direction_vec_us_2d, 0))
    return apex_mri, _unit(heading[:2])


def _nearest_voxel(idx_f):
    return np.round(idx_f).astype(int)


def mri_to_us_point(i_mri, j_mri, slice_idx, T1_vol, T1_affine, US_vol, US_affine):
    """The ultrasound voxel that shows MRI voxel (i, j, slice): through world space with both affines, rounded to the nearest
    index; returns (the ultrasound slice of constant third index through it, the index triple) -- reference
    src/cone.py:21-39, same range check and message on the MRI indices."""
    d0, d1, d2 = T1_vol.shape
    if not (0 <= slice_idx < d2 and 0 <= i_mri < d0 and 0 <= j_mri < d1):
        raise ValueError(f"T1 : indices are out of range (i={i_mri}, j={j_mri}, k={slice_idx})")
    us_idx = _nearest_voxel(world_to_voxel(voxel_to_world(np.array([i_mri, j_mri, slice_idx]), T1_affine), US_affine))
    return US_vol[:, :, us_idx[2]], us_idx


def us_to_mri_point(i_us, j_us, slice_idx, US_vol, US_affine, T1_vol, T1_affine):
    """The inverse map for an ultrasound voxel given as (slice, i, j) -- the ultrasound volume's FIRST index is the slice
    here, as in the reference --: returns (the MRI slice of constant first index through the nearest MRI voxel, the index
    triple).  Reference src/cone.py:41-59 (which does not range-check either)."""
    mri_idx = _nearest_voxel(world_to_voxel(voxel_to_world(np.array([slice_idx, i_us, j_us]), US_affine), T1_affine))
    return T1_vol[mri_idx[0], :, :], mri_idx


def rotation_from_rotvec(rotvec: torch.Tensor) -> torch.Tensor:
    """Rodrigues' formula, differentiable at 0: R = I + A K + B K^2 with K = [rotvec]x, A = sin t / t, B = (1 - cos t) / t^2,
    both from their series below 1e-3 rad (so that R and its gradient are exact for the identity)."""
    t2 = (rotvec * rotvec).sum()
    small = t2 < 1e-6
    t2s = torch.where(small, torch.ones_like(t2), t2)
    t = torch.sqrt(t2s)
    A = torch.where(small, 1.0 - t2 / 6.0, torch.sin(t) / t)
    B = torch.where(small, 0.5 - t2 / 24.0, (1.0 - torch.cos(t)) / t2s)
    x, y, z = rotvec[0], rotvec[1], rotvec[2]
    o = torch.zeros_like(x)
    K = torch.stack([torch.stack([o, -z, y]), torch.stack([z, o, -x]), torch.stack([-y, x, o])])
    return torch.eye(3, dtype=rotvec.dtype, device=rotvec.device) + A * K + B * (K @ K)


class _FanDirsFn(torch.autograd.Function):
    """(median angle, opening angle[, rotation vector]) -> directions and the adjoint, one HIP launch each way
    (diffus_fan_pose_fwd / _bwd, csrc/pose.hip).  Batched: median (P,), opening () or (P,), rotvec (P,3) or None."""

    @staticmethod
    def forward(ctx, median, opening, rotvec, n_rays, single=False):
        from . import _lib
        from .renderer import _Scope, _ptr, _stream
        lib = _lib.load()
        dev = median.device
        m = median.detach().to(torch.float32).contiguous()
        op = opening.detach().to(device=dev, dtype=torch.float32).contiguous()
        rv = None if rotvec is None else rotvec.detach().to(torch.float32).contiguous()
        P = m.numel()
        stride = 0 if op.numel() == 1 else 1
        if stride and op.numel() != P:
            raise ValueError("opening angle: one value, or one per pose")
        dirs = torch.empty((P, int(n_rays), 3), dtype=torch.float32, device=dev)
        with _Scope(dev):
            _lib.check(lib.diffus_fan_pose_fwd(_ptr(m), _ptr(op), stride, _ptr(rv), P, int(n_rays), _ptr(dirs), _stream(dev)),
                       "diffus_fan_pose_fwd")
        ctx.save_for_backward(m, op, rv)
        ctx.meta = (P, int(n_rays), stride, median.shape, opening.shape, opening.device, None if rotvec is None else rotvec.shape)
        return dirs.view(int(n_rays), 3) if single else dirs      # (one pose: (R,3) straight away -- a select afterwards is two launches in its backward)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gdirs):
        from . import _lib
        from .renderer import _Scope, _ptr, _stream
        lib = _lib.load()
        m, op, rv = ctx.saved_tensors
        P, R, stride, mshape, oshape, odev, rshape = ctx.meta
        dev = m.device
        need_m, need_o, need_r = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2] and rv is not None
        g = gdirs.to(torch.float32).contiguous()
        out = torch.empty((5, P), dtype=torch.float32, device=dev)     # rows: median, opening, rotvec x 3 (as (P,3) in rows 2..4's storage)
        g_m = out[0] if need_m else None
        g_o = out[1] if need_o else None
        g_r = out[2:].view(-1)[:3 * P].view(P, 3) if need_r else None
        with _Scope(dev):
            _lib.check(lib.diffus_fan_pose_bwd(_ptr(m), _ptr(op), stride, _ptr(rv), _ptr(g), P, R, _ptr(g_m), _ptr(g_o), _ptr(g_r),
                                               _stream(dev)), "diffus_fan_pose_bwd")
        if need_o:
            g_o = (g_o.sum() if stride == 0 else g_o).reshape(oshape).to(odev)
        return (g_m.reshape(mshape) if need_m else None, g_o, g_r.reshape(rshape) if need_r else None, None, None)


def fan_directions(median_angle: torch.Tensor, opening_angle, n_rays: int, rotvec=None) -> torch.Tensor:
    """The fan(s) of `fan_directions_torch` turned by `rotation_from_rotvec(rotvec)`, differentiable in all three, as ONE
    launch each way when the parameters live on the GPU (csrc/pose.hip; float64 inside): a scalar median angle gives
    (n_rays, 3), a (P,) one (P, n_rays, 3) with rotvec (P, 3) and the opening angle shared or (P,).  On host tensors: the
    same map composed from torch ops (what the HIP pair is tested against)."""
    median_angle = torch.as_tensor(median_angle)
    opening_angle = torch.as_tensor(opening_angle, dtype=torch.float32)
    if median_angle.device.type != "cuda":
        if median_angle.dim() == 0:
            d = fan_directions_torch(median_angle, opening_angle.to(median_angle.device), n_rays)
            return d if rotvec is None else d @ rotation_from_rotvec(rotvec).T
        op = opening_angle.to(median_angle.device).expand(median_angle.shape)
        return torch.stack([fan_directions(median_angle[p], op[p], n_rays, None if rotvec is None else rotvec[p])
                            for p in range(median_angle.shape[0])])
    single = median_angle.dim() == 0
    rv = rotvec
    if rv is not None:
        rv = torch.as_tensor(rv).to(median_angle.device)
        rv = rv.reshape(1, 3) if single else rv
    return _FanDirsFn.apply(median_angle.reshape(-1), opening_angle, rv, n_rays, single)


class FanPose(torch.nn.Module):
    """Differentiable probe pose: (apex, median angle, opening angle[, rotation vector]) -> (source, directions).

    The reference builds `source` / `directions` once with NumPy and cannot optimise them
    (SURVEY D3); the HIP backward produces d/d source and d/d directions, and this module carries
    them to the pose parameters.  forward() == (apex, generate_cone_directions((cos m, sin m),
    opening, n_rays)) up to rounding.

    One pose: apex (3,), direction (2,).  A SWEEP of P poses optimised together (the batch `render_poses` takes): apex (P,3),
    direction (P,2), rotvec (P,3) -> source (P,3), directions (P, n_rays, 3).
    """

    def __init__(self, apex, direction, opening_angle: float, n_rays: int, learn_opening: bool = False, rotvec=None):
        super().__init__()
        self.n_rays = int(n_rays)
        self.apex = torch.nn.Parameter(torch.as_tensor(apex, dtype=torch.float32).clone())
        direction = np.asarray(direction.detach().cpu() if isinstance(direction, torch.Tensor) else direction, dtype=np.float64)
        if self.apex.dim() == 1:
            med = torch.tensor(median_angle_of(direction), dtype=torch.float32)
        else:
            med = torch.tensor([median_angle_of(d) for d in direction], dtype=torch.float32)
        self.median_angle = torch.nn.Parameter(med)
        op = torch.tensor(float(opening_angle), dtype=torch.float32)
        if learn_opening:
            self.opening_angle = torch.nn.Parameter(op)
        else:
            self.register_buffer("opening_angle", op)      # a buffer: it moves with .cuda() (no upload per call, graph-capturable)
        # Six degrees of freedom: `rotvec` (axis x angle, radians; None = the reference's in-plane fan: apex + one angle)
        # turns the whole fan about its apex -- directions = R(rotvec) . fan(median angle).  Its first two components tilt the
        # fan's plane out of the slice (roll / pitch: what src/cone.py:242-258 cannot express, z = 0 there); the third is an
        # in-plane turn like the median angle.  The HIP backward's d/d directions reaches it through R.
        self.rotvec = None if rotvec is None else torch.nn.Parameter(torch.as_tensor(rotvec, dtype=torch.float32).clone())

    def forward(self):
        dirs = fan_directions(self.median_angle, self.opening_angle, self.n_rays, self.rotvec)
        if self.rotvec is None:
            # an in-plane fan by construction (dim-2 components are exact zeros): tell the renderer, which otherwise would have
            # to read a device tensor back to know -- and does not do that for tensors that require grad (renderer._fans_planar)
            dirs._diffus_planar = True
        return self.apex, dirs
