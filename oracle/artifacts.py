"""CPU restatement of the reference's artifact chain (plot_beam_frame(artifacts=True),
src/renderer.py:264-273: add_speckle_arcs_np :545-583, add_depth_dependent_lateral_blur_np :585-601,
sharpen_np :535-543).  TEST INFRASTRUCTURE ONLY.  NumPy, float64, own Gaussian filtering (SciPy's
kernel and 'reflect' boundary restated); pinned by tests/golden/g14_artifacts.npz."""
from __future__ import annotations

import numpy as np


def _reflect(j, n):
    j = np.mod(j, 2 * n)
    return np.where(j < n, j, 2 * n - 1 - j)


def gaussian_filter1d(a, sigma, axis):
    """scipy.ndimage.gaussian_filter1d(mode='reflect', truncate=4.0) restated."""
    rad = int(4.0 * sigma + 0.5)
    if rad == 0:
        return a.copy()
    x = np.arange(-rad, rad + 1)
    k = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    k /= k.sum()
    a = np.moveaxis(a, axis, 0)
    n = a.shape[0]
    out = np.zeros_like(a)
    for t, w in zip(x, k):
        out += w * a[_reflect(np.arange(n) + t, n)]
    return np.moveaxis(out, 0, axis)


def speckle(frame, std_radial, std_local, radial, local):
    noised = np.asarray(frame, np.float64) * (radial[None, :] * local)
    noised[noised < 0] = 0.0
    return noised


def noise_scales(N, std_radial, std_local):
    depth = np.linspace(0.0, 1.0, N)
    return std_radial * (1.0 + depth ** 2.0), std_local * (1.0 + depth ** 1.5)


def lateral_blur(img, max_sigma):
    R, N = img.shape
    out = img.copy()
    for z in range(N):
        sigma = max_sigma * (z / (N - 1)) if z > 0 else 1e-8
        out[:, z] = gaussian_filter1d(out[:, z], sigma, 0)
    return out


def sharpen(img, alpha):
    blurred = gaussian_filter1d(gaussian_filter1d(img, 1.0, 0), 1.0, 1)
    return np.clip(img + alpha * (img - blurred), img.min(), img.max())


def chain(frame, std_radial, std_local, max_sigma, alpha, radial, local):
    return sharpen(lateral_blur(speckle(frame, std_radial, std_local, radial, local), max_sigma), alpha)
