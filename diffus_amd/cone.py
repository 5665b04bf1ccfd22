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
    """Generate a fan of directions centered on direction_mri_world, spanning
    opening_angle (radians), in the (x, y) plane (z=0).  Returns (n_rays, 3) float32.
    Same signature and values as reference src/cone.py:242."""
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


# ---- probe-pose helpers of the reference (host-side NumPy geometry) -----------
def voxel_to_world(idx_ijk, affine):
    """reference src/cone.py:10-13"""
    ijk1 = np.concatenate((idx_ijk, [1.0]))
    return affine.dot(ijk1)[:3]


def world_to_voxel(xyz, affine):
    """reference src/cone.py:15-19"""
    xyz1 = np.concatenate((xyz, [1.0]))
    return np.linalg.inv(affine).dot(xyz1)[:3]


def compute_us_apex_and_direction(m_left, b_left, m_right, b_right):
    """Apex, opening angle and bisector of the fan bounded by the two edge lines
    y = m_left x + b_left and y = m_right x + b_right (reference src/cone.py:98-126)."""
    if np.isclose(m_left, m_right):
        raise RuntimeError("The slopes are nearly equal; no defined intersection.")
    x0 = (b_right - b_left) / (m_left - m_right)
    y0 = m_left * x0 + b_left
    v_left = np.array([-1, -m_left])
    v_right = np.array([1, m_right])
    u_left = v_left / np.linalg.norm(v_left)
    u_right = v_right / np.linalg.norm(v_right)
    opening_angle = np.arccos(np.clip(np.dot(u_left, u_right), -1.0, 1.0))
    bisector = u_left + u_right
    bisector = bisector / np.linalg.norm(bisector)
    return {"apex": (x0, y0), "opening_angle": opening_angle, "direction_vector": bisector}


def cone_us_to_mri_world(apex_us_vox, direction_vec_us_2d, US_affine, T1_affine):
    """US-voxel apex and in-plane direction -> MRI voxel apex and unit in-plane direction
    (reference src/cone.py:187-209)."""
    apex_t1_vox = world_to_voxel(voxel_to_world(apex_us_vox, US_affine), T1_affine)
    direction_vec_3d = np.append(direction_vec_us_2d, 0)
    direction_vec_t1 = T1_affine[:3, :3] @ (np.linalg.inv(US_affine[:3, :3]) @ direction_vec_3d)
    return apex_t1_vox, direction_vec_t1[:2] / np.linalg.norm(direction_vec_t1[:2])


class FanPose(torch.nn.Module):
    """Differentiable probe pose: (apex, median angle, opening angle) -> (source, directions).

    The reference builds `source` / `directions` once with NumPy and cannot optimise them
    (SURVEY D3); the HIP backward produces d/d source and d/d directions, and this module carries
    them to the three pose parameters.  forward() == (apex, generate_cone_directions((cos m, sin m),
    opening, n_rays)) up to rounding.
    """

    def __init__(self, apex, direction, opening_angle: float, n_rays: int, learn_opening: bool = False):
        super().__init__()
        self.n_rays = int(n_rays)
        self.apex = torch.nn.Parameter(torch.as_tensor(apex, dtype=torch.float32).clone())
        self.median_angle = torch.nn.Parameter(torch.tensor(median_angle_of(direction), dtype=torch.float32))
        op = torch.tensor(float(opening_angle), dtype=torch.float32)
        self.opening_angle = torch.nn.Parameter(op) if learn_opening else op

    def forward(self):
        op = self.opening_angle.to(self.median_angle.device)
        return self.apex, fan_directions_torch(self.median_angle, op, self.n_rays)
