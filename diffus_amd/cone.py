"""Fan geometry: mirror of the reference's src/cone.py:242-258.

`generate_cone_directions` is bit-exact with the reference (tests/golden G8).
`fan_directions_torch` is its differentiable torch twin, so that the pose
gradients produced by the HIP backward reach (median angle, opening angle).
"""
from __future__ import annotations

import math

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
