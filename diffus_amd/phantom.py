"""Synthetic inputs for benchmarks and tests.

No impedance volume ships with the reference (its .nii.gz files are
git-ignored), so every config runs on this analytic phantom.  It is
pure NumPy with closed-form expressions -- no RNG -- so the build container,
the GPU box and the oracle all regenerate bit-identical inputs.
"""
from __future__ import annotations

import math

import numpy as np

AIR = 400.0          # src/impedance.py:52 of the reference uses 400 for air
BONE = 6.4e6         # bone-scale impedance (cf. USPhysics.md:54)
CSF = 1.50e6         # values from `[DEMO] Modeling Choices` cell 5
BRAIN = 1.60e6
TUMOUR = 1.68e6


def phantom(n: int, variant: int = 0) -> np.ndarray:
    """n^3 float32 head phantom, indexed [dim0, dim1, dim2] like the reference.

    Voxel centre u = index/(n-1) in [0,1]^3.  Air background; a 0.025-thick
    skull shell on the ellipsoid with semi-axes (0.46,0.40,0.44); CSF; brain
    inside (0.42,0.36,0.40); a tumour sphere r=0.08 at (0.58,0.45,0.52) shifted
    by 0.02*variant along dim 0; plus a smooth texture inside the head so that
    every in-tissue step sees a non-zero reflection coefficient.
    """
    u = np.arange(n, dtype=np.float64) / (n - 1)
    c = u - 0.5

    def ell(a0, a1, a2):
        return ((c / a0) ** 2)[:, None, None] + ((c / a1) ** 2)[None, :, None] + ((c / a2) ** 2)[None, None, :]

    vol = np.full((n, n, n), AIR, dtype=np.float32)
    outer = ell(0.46, 0.40, 0.44) <= 1.0
    vol[outer] = BONE
    inner = ell(0.46 - 0.025, 0.40 - 0.025, 0.44 - 0.025) <= 1.0
    vol[inner] = CSF
    del inner
    brain = ell(0.42, 0.36, 0.40) <= 1.0
    vol[brain] = BRAIN
    del brain
    tc = (0.58 + 0.02 * variant, 0.45, 0.52)
    tum = (((u - tc[0]) ** 2)[:, None, None] + ((u - tc[1]) ** 2)[None, :, None]
           + ((u - tc[2]) ** 2)[None, None, :]) <= 0.08 ** 2
    vol[tum] = TUMOUR
    del tum
    tex = (2e4 * np.sin(37 * u)[:, None, None] * np.sin(29 * u)[None, :, None]
           * np.sin(31 * u)[None, None, :]).astype(np.float32)
    vol += np.where(outer, tex, np.float32(0))
    return vol


def cone_directions_np(direction, opening_angle: float, n_rays: int) -> np.ndarray:
    """NumPy core of generate_cone_directions (reference src/cone.py:242-258):
    n_rays unit vectors in the (0,1) plane spanning opening_angle (radians)
    about the normalised first two components of `direction`; dim-2 component 0.
    Computed in float64, returned as float32 (the reference builds a float32
    tensor from float64 rows)."""
    d = np.array(direction[:2])          # keeps the caller's dtype, like the reference (float32 tensors
    d = d / np.linalg.norm(d)            # are normalised in float32 there)
    ortho = np.array([-d[1], d[0]])
    ang = np.linspace(-opening_angle / 2, opening_angle / 2, n_rays)
    v = np.cos(ang)[:, None] * d[None, :] + np.sin(ang)[:, None] * ortho[None, :]
    out = np.zeros((n_rays, 3), dtype=np.float64)
    out[:, :2] = v
    return out.astype(np.float32)


def pose_ring(n: int, P: int, R: int, opening_deg: float = 60.0, phase: float = 0.0, roll_deg: float = 0.0,
              pitch_deg: float = 0.0, plane=(0, 1)):
    """P probe poses on a ring inside the head; `phase` (radians) turns the whole ring.

    -> sources (P,3) float32, directions (P,R,3) float32.  Apex p sits at
    (0.5n + 0.30n cos phi, 0.5n + 0.30n sin phi, 0.5n + 0.05n sin 3phi) and the
    fan looks at the volume centre, in the (0,1) plane like every demo fan.

    Fans a pose optimisation produces (`plot_beam_frame` takes any `directions`, src/renderer.py:119-124): `roll_deg`
    turns each fan's plane about its central ray (the plane's normal leaves dim 2 by that angle; the edge rays climb
    along dim 2, the central ray does not), `pitch_deg` lifts the central ray itself out of the slice.  `plane=(a, b)`
    puts the ring and the fans into the plane of dims (a, b) instead of (0, 1) -- a permutation of the coordinates.
    Unit directions up to float32 rounding, computed in float64.
    """
    src = np.zeros((P, 3), dtype=np.float32)
    dirs = np.zeros((P, R, 3), dtype=np.float32)
    roll, pitch = math.radians(roll_deg), math.radians(pitch_deg)
    a, b = plane
    c = 3 - a - b
    perm = [0, 0, 0]
    perm[a], perm[b], perm[c] = 0, 1, 2
    for p in range(P):
        phi = 2.0 * math.pi * p / P + phase
        s = np.array((0.5 * n + 0.30 * n * math.cos(phi),
                      0.5 * n + 0.30 * n * math.sin(phi),
                      0.5 * n + 0.05 * n * math.sin(3 * phi)))
        d = cone_directions_np((-math.cos(phi), -math.sin(phi)), math.radians(opening_deg), R)
        if roll_deg != 0.0 or pitch_deg != 0.0:
            look = np.array((-math.cos(phi), -math.sin(phi), 0.0))
            side = np.array((math.sin(phi), -math.cos(phi), 0.0))
            up = np.array((0.0, 0.0, 1.0))
            look2 = math.cos(pitch) * look + math.sin(pitch) * up
            up2 = math.cos(pitch) * up - math.sin(pitch) * look
            side2 = math.cos(roll) * side + math.sin(roll) * up2
            d64 = d.astype(np.float64)
            along, across = d64 @ look, d64 @ side                      # the flat fan in its own frame
            d = (along[:, None] * look2[None, :] + across[:, None] * side2[None, :]).astype(np.float32)
        src[p] = s[perm].astype(np.float32)
        dirs[p] = d[:, perm]
    return src, dirs
