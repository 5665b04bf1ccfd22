"""ctypes front-end of the CPU oracle (oracle/diffus_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, bench.py's cpu_baseline leg
and __graft_entry__.smoke(); never from diffus_amd/ (the product path has no
CPU fallback and fails loudly without its HIP library).

Every function is the NumPy-facing twin of the C function of the same name;
the reference lines each one restates are cited in diffus_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIFFUS_ORACLE_SO: another build of the same source (oracle/Makefile `sanitize`, tools/oracle_sanitize.sh)
_SO = os.environ.get("DIFFUS_ORACLE_SO") or os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (seconds).  Idempotent."""
    src = os.path.join(_HERE, "diffus_oracle.c")
    if os.environ.get("DIFFUS_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_start_crop.restype = C.c_int
        _lib.orc_prop_single_ray_dense.restype = C.c_int
        _lib.orc_plot_beam_frame.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _pose(source, directions):
    """Mirror torch's promotion: integer inputs behave as f32."""
    source = np.asarray(source)
    directions = np.asarray(directions)
    if source.dtype != np.float64:
        source = source.astype(np.float32)
    if directions.dtype != np.float64:
        directions = directions.astype(np.float32)
    source = np.ascontiguousarray(source).reshape(3)
    directions = np.ascontiguousarray(directions).reshape(-1, 3)
    return source, int(source.dtype == np.float64), directions, int(directions.dtype == np.float64)


def sample_nearest(vol, source, directions, S):
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    src, sf, dirs, df = _pose(source, directions)
    R = dirs.shape[0]
    ix = np.empty((R, S), np.int64); iy = np.empty((R, S), np.int64); iz = np.empty((R, S), np.int64)
    imp = np.empty((R, S), np.float32)
    lib().orc_sample_nearest(_p(vol), *map(C.c_int, vol.shape), _p(src), C.c_int(sf), _p(dirs), C.c_int(df),
                             C.c_int(R), C.c_int(S), _p(ix), _p(iy), _p(iz), _p(imp))
    return ix, iy, iz, imp


def sample_trilinear(vol, source, directions, S, want_grad=False):
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    src, sf, dirs, df = _pose(source, directions)
    R = dirs.shape[0]
    imp = np.empty((R, S), np.float32)
    g = np.empty((R, S, 3), np.float32) if want_grad else None
    lib().orc_sample_trilinear(_p(vol), *map(C.c_int, vol.shape), _p(src), C.c_int(sf), _p(dirs), C.c_int(df),
                               C.c_int(R), C.c_int(S), _p(imp), _p(g))
    return (imp, g) if want_grad else imp


def reflection(imp):
    imp = np.ascontiguousarray(imp, dtype=np.float32)
    R, S = imp.shape
    r = np.empty((R, S - 1), np.float32)
    lib().orc_reflection(_p(imp), C.c_int(R), C.c_int(S), _p(r))
    return r


def start_crop(r, start):
    """-> (cropped r, median value, index of the median ray or -1)."""
    r = np.ascontiguousarray(r, dtype=np.float32)
    R, Sm1 = r.shape
    out = np.empty((R, Sm1 - start), np.float32)
    med = C.c_float(0)
    who = lib().orc_start_crop(_p(r), C.c_int(R), C.c_int(Sm1), C.c_int(start), _p(out), C.byref(med))
    return out, med.value, who


def echo_scan(r, dtype=np.float32):
    """O(N) echo series: r (B,N) -> echo (B,N+1)."""
    r = np.ascontiguousarray(r, dtype=dtype)
    B, N = r.shape
    e = np.empty((B, N + 1), dtype)
    fn = lib().orc_echo_scan_f32 if dtype == np.float32 else lib().orc_echo_scan_f64
    fn(_p(r), C.c_int(B), C.c_int(N), _p(e))
    return e


def prop_single_ray_dense(r):
    """Literal dense solve for one ray (fp64): r (n,) -> w (2n+2,)."""
    r = np.ascontiguousarray(r, dtype=np.float64).reshape(-1)
    w = np.zeros(2 * (r.size + 1), np.float64)
    lib().orc_prop_single_ray_dense(_p(r), C.c_int(r.size), _p(w))
    return w


def sample_points_nearest(vol, pts):
    """custom_nearest_sampler (src/renderer.py:751-759) at arbitrary points: cast to float32, round half to even,
    clamp, gather.  vol (d0,d1,d2), pts (...,3) -> x, y, z (int64) and values, each of shape pts.shape[:-1]."""
    vol = np.asarray(vol)
    p = np.asarray(pts, dtype=np.float32)
    idx = [np.clip(np.rint(p[..., c]).astype(np.int64), 0, vol.shape[c] - 1) for c in range(3)]
    return idx[0], idx[1], idx[2], vol[idx[0], idx[1], idx[2]]


def attenuate(echo, alpha):
    echo = np.ascontiguousarray(echo, dtype=np.float32)
    B, N1 = echo.shape
    f = np.empty_like(echo)
    lib().orc_attenuate(_p(echo), C.c_int(B), C.c_int(N1), C.c_float(alpha), _p(f))
    return f


def resolve_start(start, S):
    """src/renderer.py:237-240: a Python float is a fraction of num_samples."""
    if type(start) is float:
        start = int(start * S)
    if type(start) is int:
        start = max(0, start)
    return start


def plot_beam_frame(vol, source, directions, S, alpha, start=0, sampler="nearest"):
    """Whole path, one pose.  -> (x, y, z, frame) cropped like src/renderer.py:275."""
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    src, sf, dirs, df = _pose(source, directions)
    start = resolve_start(start, S)
    R = dirs.shape[0]
    ix = np.empty((R, S), np.int64); iy = np.empty((R, S), np.int64); iz = np.empty((R, S), np.int64)
    frame = np.empty((R, S - start), np.float32)
    lib().orc_plot_beam_frame(_p(vol), *map(C.c_int, vol.shape), _p(src), C.c_int(sf), _p(dirs), C.c_int(df),
                              C.c_int(R), C.c_int(S), C.c_int(start), C.c_float(alpha),
                              C.c_int({"nearest": 0, "trilinear": 1}[sampler]),
                              _p(frame), _p(ix), _p(iy), _p(iz))
    return ix[:, start:], iy[:, start:], iz[:, start:], frame
