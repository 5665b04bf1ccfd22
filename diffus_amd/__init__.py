"""diffus_amd -- MI355X-native implementation of the DiffUS `plot_beam_frame` hot path.

Drop-in for `from src.renderer import *` / `from src.cone import generate_cone_directions`
of gduguey/DiffUS on that path.
"""
from ._lib import DiffusError, LIB_PATH  # noqa: F401
from .cone import (FanPose, compute_us_apex_and_direction, cone_us_to_mri_world, fan_directions, fan_directions_torch,  # noqa: F401
                   generate_cone_directions, mri_to_us_point, rotation_from_rotvec, us_to_mri_point, voxel_to_world, world_to_voxel)
from .renderer import (BrickedVolume, UltrasoundRenderer, brick_volume, compute_echo_traces,  # noqa: F401
                       compute_gaussian_pulse, custom_nearest_sampler, gaussian_pulse, prop_single_ray,
                       propagate_full_rays_batched,
                       pair_volume, render_poses, resolve_start, trace_rays, unbrick_volume)

from .artifacts import apply_artifacts  # noqa: F401,E402
from .splat import differentiable_splat, rotate_around_apex, splat_frames  # noqa: F401,E402
from .impedance import ImpedanceEstimator, create_brain_mask, masked_stats, zscore_normalize  # noqa: F401,E402
from .captured import CapturedStep  # noqa: F401,E402
from .losses import ssim_loss  # noqa: F401,E402
from .raster import rasterize_fan  # noqa: F401,E402

__all__ = ["ssim_loss", "rasterize_fan", "prop_single_ray", "propagate_full_rays_batched", "custom_nearest_sampler", "CapturedStep", "ImpedanceEstimator", "create_brain_mask", "zscore_normalize", "masked_stats", "apply_artifacts", "compute_gaussian_pulse", "gaussian_pulse", "FanPose", "compute_us_apex_and_direction", "cone_us_to_mri_world", "voxel_to_world", "world_to_voxel", "mri_to_us_point", "us_to_mri_point", "rotation_from_rotvec",
           "differentiable_splat", "rotate_around_apex", "splat_frames", "UltrasoundRenderer", "compute_echo_traces", "render_poses", "trace_rays", "resolve_start",
           "generate_cone_directions", "fan_directions", "fan_directions_torch", "DiffusError", "BrickedVolume", "brick_volume",
           "unbrick_volume", "pair_volume"]
