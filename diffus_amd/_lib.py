"""ctypes binding of libdiffus_hip.so (include/diffus_hip.h).

There is NO CPU fallback: if the library is missing or a call fails the product
path raises.  (The CPU restatement lives in oracle/ and is test-only.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIFFUS_LIB") or os.path.join(_HERE, "libdiffus_hip.so")

EXPORTS = ("diffus_abi_version", "diffus_strerror", "diffus_workspace_bytes", "diffus_workspace_zbar_offset",
           "diffus_bricked_floats", "diffus_brick_volume", "diffus_unbrick_volume", "diffus_paired_floats",
           "diffus_pair_volume", "diffus_convert_volume_box", "diffus_brick_count", "diffus_gradbuf_flush",
           "diffus_render_fwd", "diffus_render_bwd", "diffus_render_bwd_mse", "diffus_render_step_mse", "diffus_trace_rays", "diffus_echo_traces",
           "diffus_loss_sumsq", "diffus_splat_workspace_bytes", "diffus_splat_fwd", "diffus_splat_bwd",
           "diffus_artifacts_workspace_bytes", "diffus_artifacts",
           "diffus_mlp_fwd", "diffus_mlp_workspace_bytes", "diffus_mlp_bwd", "diffus_brain_mask_workspace_bytes",
           "diffus_brain_mask", "diffus_masked_stats_workspace_bytes", "diffus_masked_stats", "diffus_rows_conv1d",
           "diffus_prop_single_ray", "diffus_propagate_rays", "diffus_sample_points",
           "diffus_echo_bwd_workspace_bytes", "diffus_echo_traces_bwd", "diffus_splat_axes", "diffus_rotate_around_apex",
           "diffus_ssim_workspace_bytes", "diffus_ssim_loss_fwd", "diffus_ssim_loss_bwd", "diffus_fan_pose_fwd", "diffus_fan_pose_bwd")

ABI_VERSION = 8          # include/diffus_hip.h DIFFUS_ABI_VERSION
DIFFUS_F32, DIFFUS_F64, DIFFUS_I64 = 0, 1, 2
NEAREST, TRILINEAR = 0, 1
CANONICAL, BRICKED, PAIRED = 0, 1, 2
GRAD_BRICKED = 0x10   # OR'ed into `layout` of the backward calls: the gradient is the bricked scratch whatever the volume's layout
FANS_PLANAR = 0x20    # OR'ed into `layout` of the backward calls: a hint that no ray moves along dim 2 (the scatter launch for planar fans)
FLUSH_STORE, FLUSH_ACCUMULATE, FLUSH_PERSISTENT, FLUSH_DENSE = 0, 1, 2, 3   # diffus_gradbuf_flush modes
BWD_SCAN, BWD_SCATTER, BWD_ALL = 1, 2, 3
BWD_KEEP_MEDIAN = 4         # start > 0: the workspace still holds the forward's median (include/diffus_hip.h)
BWD_REPAIR_FRAME = 8        # one-pass step: float64 frame rows for ill-conditioned rays (include/diffus_hip.h; ~10 us per step)
MAX_SAMPLES = 1024          # cropped samples per launch; longer rays run as chained segments
MAX_SEGMENTS = 64

_lib = None


class DiffusError(RuntimeError):
    pass


def load():
    """Load the HIP library; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DiffusError(
            f"{LIB_PATH} not found: build it with `python -m diffus_amd.build` "
            "(hipcc --offload-arch=gfx950).  diffus_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.diffus_abi_version.restype = i
    lib.diffus_strerror.restype = C.c_char_p
    lib.diffus_strerror.argtypes = [i]
    lib.diffus_workspace_bytes.restype = sz
    lib.diffus_workspace_bytes.argtypes = [i, i, i, i]
    lib.diffus_workspace_zbar_offset.restype = sz
    lib.diffus_workspace_zbar_offset.argtypes = [i, i, i, i]
    common = [vp, i, i, i, i, vp, i, vp, i, i, i, i, i, f, i]
    lib.diffus_render_fwd.restype = i
    lib.diffus_render_fwd.argtypes = common + [vp, vp, vp, sz, vp]
    lib.diffus_render_bwd.restype = i
    lib.diffus_render_bwd.argtypes = common + [vp, vp, vp, vp, vp, i, vp, sz, vp]
    lib.diffus_render_bwd_mse.restype = i
    lib.diffus_render_bwd_mse.argtypes = common + [vp, vp, f, vp, vp, vp, vp, vp, i, vp, sz, vp]
    lib.diffus_render_step_mse.restype = i
    lib.diffus_render_step_mse.argtypes = common + [vp, f, vp, vp, vp, vp, vp, vp, i, vp, sz, vp]
    lib.diffus_brick_count.restype = sz
    lib.diffus_brick_count.argtypes = [i, i, i]
    lib.diffus_gradbuf_flush.restype = i
    lib.diffus_gradbuf_flush.argtypes = [vp, vp, i, i, i, vp, i, vp]
    lib.diffus_loss_sumsq.restype = i
    lib.diffus_loss_sumsq.argtypes = [vp, i, C.c_long, vp, vp, vp, sz, vp]
    lib.diffus_trace_rays.restype = i
    lib.diffus_trace_rays.argtypes = [vp, i, i, i, i, vp, i, vp, i, i, i, i, i, vp, vp, vp, vp]
    lib.diffus_bricked_floats.restype = sz
    lib.diffus_bricked_floats.argtypes = [i, i, i]
    lib.diffus_paired_floats.restype = sz
    lib.diffus_paired_floats.argtypes = [i, i, i]
    lib.diffus_pair_volume.restype = i
    lib.diffus_pair_volume.argtypes = [vp, i, i, i, vp, vp]
    lib.diffus_convert_volume_box.restype = i
    lib.diffus_convert_volume_box.argtypes = [vp, i, i, i, i, vp, i, i, i, i, i, i, vp]
    lib.diffus_brick_volume.restype = i
    lib.diffus_brick_volume.argtypes = [vp, i, i, i, vp, vp]
    lib.diffus_unbrick_volume.restype = i
    lib.diffus_unbrick_volume.argtypes = [vp, i, i, i, vp, i, vp]
    lib.diffus_echo_traces.restype = i
    lib.diffus_echo_traces.argtypes = [vp, i, i, vp, vp]
    lib.diffus_echo_bwd_workspace_bytes.restype = sz
    lib.diffus_echo_bwd_workspace_bytes.argtypes = [i, i]
    lib.diffus_echo_traces_bwd.restype = i
    lib.diffus_echo_traces_bwd.argtypes = [vp, i, i, vp, vp, vp, sz, vp]
    lib.diffus_splat_workspace_bytes.restype = sz
    lib.diffus_splat_workspace_bytes.argtypes = [i, i, i]
    lib.diffus_ssim_workspace_bytes.restype = sz
    lib.diffus_ssim_workspace_bytes.argtypes = [i, i, i]
    lib.diffus_ssim_loss_fwd.restype = i
    lib.diffus_ssim_loss_fwd.argtypes = [vp, vp, i, i, i, i, f, f, f, vp, vp, sz, vp]
    lib.diffus_ssim_loss_bwd.restype = i
    lib.diffus_ssim_loss_bwd.argtypes = [vp, vp, i, i, i, i, f, f, f, vp, vp, i, vp, sz, vp]
    lib.diffus_rotate_around_apex.restype = i
    lib.diffus_rotate_around_apex.argtypes = [vp, vp, C.c_long, vp, vp, f, vp, vp, vp]
    lib.diffus_fan_pose_fwd.restype = i
    lib.diffus_fan_pose_fwd.argtypes = [vp, vp, i, vp, i, i, vp, vp]
    lib.diffus_fan_pose_bwd.restype = i
    lib.diffus_fan_pose_bwd.argtypes = [vp, vp, i, vp, vp, i, i, vp, vp, vp, vp]
    lib.diffus_splat_axes.restype = i
    lib.diffus_splat_axes.argtypes = [vp, i, vp, i, vp, i, C.c_long, vp, vp, vp, vp]
    lib.diffus_splat_fwd.restype = i
    lib.diffus_splat_fwd.argtypes = [vp, vp, vp, i, C.c_long, i, i, i, f, vp, vp, vp, sz, vp]
    lib.diffus_splat_bwd.restype = i
    lib.diffus_splat_bwd.argtypes = [vp, vp, i, C.c_long, i, i, f, vp, vp, vp, vp, sz, vp]
    d = C.c_double
    lib.diffus_artifacts_workspace_bytes.restype = sz
    lib.diffus_artifacts_workspace_bytes.argtypes = [i, i, i]
    lib.diffus_artifacts.restype = i
    lib.diffus_artifacts.argtypes = [vp, i, i, i, d, d, d, d, vp, vp, C.c_uint64, vp, vp, sz, vp]
    lib.diffus_mlp_fwd.restype = i
    lib.diffus_mlp_fwd.argtypes = [vp, vp, sz, vp, f, f, f, f, vp, sz, vp]
    lib.diffus_mlp_workspace_bytes.restype = sz
    lib.diffus_mlp_workspace_bytes.argtypes = []
    lib.diffus_mlp_bwd.restype = i
    lib.diffus_mlp_bwd.argtypes = [vp, vp, sz, vp, f, f, f, vp, sz, vp, vp, vp, sz, vp]
    lib.diffus_brain_mask_workspace_bytes.restype = sz
    lib.diffus_brain_mask_workspace_bytes.argtypes = [i, i, i]
    lib.diffus_brain_mask.restype = i
    lib.diffus_brain_mask.argtypes = [vp, i, i, i, f, i, vp, vp, sz, vp]
    lib.diffus_masked_stats_workspace_bytes.restype = sz
    lib.diffus_masked_stats_workspace_bytes.argtypes = []
    lib.diffus_masked_stats.restype = i
    lib.diffus_masked_stats.argtypes = [vp, vp, sz, vp, vp, sz, vp]
    lib.diffus_rows_conv1d.restype = i
    lib.diffus_rows_conv1d.argtypes = [vp, i, i, vp, i, i, vp, vp]
    lib.diffus_prop_single_ray.restype = i
    lib.diffus_prop_single_ray.argtypes = [vp, i, i, i, vp, vp]
    lib.diffus_propagate_rays.restype = i
    lib.diffus_propagate_rays.argtypes = [vp, i, i, vp, vp]
    lib.diffus_sample_points.restype = i
    lib.diffus_sample_points.argtypes = [vp, i, i, i, i, vp, C.c_long, i, vp, vp, vp]
    if lib.diffus_abi_version() != ABI_VERSION:
        raise DiffusError("libdiffus_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        raise DiffusError(f"{what} failed: {load().diffus_strerror(rc).decode()} ({rc})")
