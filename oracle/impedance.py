"""CPU restatement of the reference's MRI -> impedance stage.  TEST INFRASTRUCTURE ONLY.

  ImpedanceEstimator.forward          src/impedance.py:6-17   1 -> 32 -> 32 -> 1 MLP, ReLU
  create_brain_mask                   src/utils.py:12-21      threshold, binary dilation x2, binary erosion x2
  zscore_normalize                    src/utils.py:23-39      (v - mean) / (std + 1e-8) over the masked voxels
  ImpedanceEstimator.compute_impedance_volume  src/impedance.py:38-53

NumPy only; the SciPy morphology is restated with array shifts (6-neighbourhood, outside = 0).
Pinned by tests/golden/g15_impedance.npz (outputs of the reference itself, tests/golden/make_golden.py).
"""
from __future__ import annotations

import numpy as np

HIDDEN = 32
NPARAMS = 32 + 32 + 32 * 32 + 32 + 32 + 1       # W1 b1 W2 b2 W3 b3, the order of the torch state_dict


def pack(sd, prefix="model."):
    """state_dict arrays -> flat float32 vector [W1(32) b1(32) W2(32x32, out-major) b2(32) W3(32) b3(1)]."""
    parts = [sd[prefix + "0.weight"].reshape(-1), sd[prefix + "0.bias"], sd[prefix + "2.weight"].reshape(-1),
             sd[prefix + "2.bias"], sd[prefix + "4.weight"].reshape(-1), sd[prefix + "4.bias"]]
    v = np.concatenate([np.asarray(p, np.float32) for p in parts])
    assert v.size == NPARAMS
    return v


def unpack(p):
    p = np.asarray(p)
    return p[0:32], p[32:64], p[64:1088].reshape(32, 32), p[1088:1120], p[1120:1152], p[1152]


def mlp_forward(x, params, dtype=np.float32):
    """x (...,) -> y (...,): W3 relu(W2 relu(W1 x + b1) + b2) + b3."""
    W1, b1, W2, b2, W3, b3 = [np.asarray(a, dtype) for a in unpack(params)]
    xf = np.asarray(x, dtype).reshape(-1, 1)
    h1 = np.maximum(xf * W1[None, :] + b1[None, :], 0)
    h2 = np.maximum(h1 @ W2.T + b2[None, :], 0)
    return (h2 @ W3 + b3).reshape(np.shape(x))


def mlp_backward(x, params, gy, dtype=np.float64):
    """-> (gparams flat, gx) of sum(y * gy)."""
    W1, b1, W2, b2, W3, b3 = [np.asarray(a, dtype) for a in unpack(params)]
    xf = np.asarray(x, dtype).reshape(-1, 1)
    g = np.asarray(gy, dtype).reshape(-1, 1)
    a1 = xf * W1[None, :] + b1[None, :]
    h1 = np.maximum(a1, 0)
    a2 = h1 @ W2.T + b2[None, :]
    h2 = np.maximum(a2, 0)
    g2 = (g * W3[None, :]) * (a2 > 0)
    g1 = (g2 @ W2) * (a1 > 0)
    gp = np.concatenate([(g1 * xf).sum(0), g1.sum(0), (g2.T @ h1).reshape(-1), g2.sum(0), (h2 * g).sum(0), [g.sum()]])
    return gp, (g1 @ W1).reshape(np.shape(x))


def _shift(a, axis, d):
    """a shifted by d along axis, zeros entering (SciPy border_value = 0)."""
    out = np.zeros_like(a)
    src = [slice(None)] * a.ndim
    dst = [slice(None)] * a.ndim
    if d > 0:
        src[axis] = slice(0, -d); dst[axis] = slice(d, None)
    else:
        src[axis] = slice(-d, None); dst[axis] = slice(0, d)
    out[tuple(dst)] = a[tuple(src)]
    return out


def binary_dilation(m, iterations):
    m = np.asarray(m, bool)
    for _ in range(iterations):
        o = m.copy()
        for ax in range(m.ndim):
            o |= _shift(m, ax, 1) | _shift(m, ax, -1)
        m = o
    return m


def binary_erosion(m, iterations):
    m = np.asarray(m, bool)
    for _ in range(iterations):
        o = m.copy()
        for ax in range(m.ndim):
            o &= _shift(m, ax, 1) & _shift(m, ax, -1)
        m = o
    return m


def create_brain_mask(volume, threshold=50, iterations=2):
    return binary_erosion(binary_dilation(np.asarray(volume) > threshold, iterations), iterations)


def masked_stats(volume, mask):
    """mean and unbiased std of the masked voxels (torch .mean() / .std())."""
    v = np.asarray(volume, np.float64)[np.asarray(mask, bool)]
    return v.mean(), v.std(ddof=1)


def zscore_normalize(volume, mask):
    mean, std = masked_stats(volume, mask)
    v = np.asarray(volume, np.float32)
    return (v - np.float32(mean)) / (np.float32(std) + np.float32(1e-8))


def compute_impedance_volume(volume, params, threshold=50):
    mask = create_brain_mask(volume, threshold)
    vn = zscore_normalize(volume, mask)
    Z = np.full(np.shape(volume), 400.0, np.float32)
    Z[mask] = mlp_forward(vn[mask], params) * np.float32(1e6)
    return Z, mask
