"""diffus_amd -- MI355X-native implementation of the DiffUS `plot_beam_frame` hot path.

Drop-in for `from src.renderer import *` / `from src.cone import generate_cone_directions`
of gduguey/DiffUS on that path.
"""
from ._lib import DiffusError, LIB_PATH  # noqa: F401
from .cone import fan_directions_torch, generate_cone_directions  # noqa: F401
from .renderer import (BrickedVolume, UltrasoundRenderer, brick_volume, compute_echo_traces,  # noqa: F401
                       render_poses, resolve_start, trace_rays, unbrick_volume)

from .splat import differentiable_splat, rotate_around_apex, splat_frames  # noqa: F401,E402

__all__ = ["differentiable_splat", "rotate_around_apex", "splat_frames", "UltrasoundRenderer", "compute_echo_traces", "render_poses", "trace_rays", "resolve_start",
           "generate_cone_directions", "fan_directions_torch", "DiffusError", "BrickedVolume", "brick_volume",
           "unbrick_volume"]
