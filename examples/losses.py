"""Losses the reference's notebooks attach AFTER the hot path -- plain torch, NOT part of diffus_amd.

`ssim`: the structural similarity index of Wang, Bovik, Sheikh & Simoncelli (IEEE TIP 2004) as the notebook
`[DEMO] Train MRI to Impedance MLP - GPU` cell 16 calls it: `piq.ssim(x, y, data_range=1.0)` with piq's defaults --
11 x 11 Gaussian window of sigma 1.5 (normalised, 'valid' convolution), K1 = 0.01, K2 = 0.03, mean over the map.
piq (photosynthesis-team/piq) is a third-party package the reference imports in that notebook only; it is neither in
the reference's requirements.txt nor installed here, so this is a restatement of the published formula, parity
UNPINNED (no golden vector of piq's exists in the reference).  For inputs of 256 x 256 piq's optional average-pool
down-sampling (factor max(1, round(min(H, W) / 256))) is the identity.
"""
import torch
import torch.nn.functional as F


def gaussian_window(size: int = 11, sigma: float = 1.5, device=None) -> torch.Tensor:
    c = torch.arange(size, dtype=torch.float32, device=device) - (size - 1) / 2.0
    g = torch.exp(-(c[:, None] ** 2 + c[None, :] ** 2) / (2.0 * sigma ** 2))
    return (g / g.sum())[None, None]


def ssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0, window: torch.Tensor = None, k1: float = 0.01,
         k2: float = 0.03) -> torch.Tensor:
    """x, y: (N,1,H,W) in [0, data_range] -> scalar mean SSIM."""
    w = gaussian_window(device=x.device) if window is None else window
    f = max(1, round(min(x.shape[-2:]) / 256))
    if f > 1:
        x, y = F.avg_pool2d(x, f), F.avg_pool2d(y, f)
    x, y = x / data_range, y / data_range
    c1, c2 = k1 ** 2, k2 ** 2
    mu_x, mu_y = F.conv2d(x, w), F.conv2d(y, w)
    mu_xx, mu_yy, mu_xy = mu_x * mu_x, mu_y * mu_y, mu_x * mu_y
    s_xx = F.conv2d(x * x, w) - mu_xx
    s_yy = F.conv2d(y * y, w) - mu_yy
    s_xy = F.conv2d(x * y, w) - mu_xy
    cs = (2.0 * s_xy + c2) / (s_xx + s_yy + c2)
    ss = (2.0 * mu_xy + c1) / (mu_xx + mu_yy + c1) * cs
    return ss.mean()


def minmax01(img: torch.Tensor) -> torch.Tensor:
    """(img - min) / (max - min + 1e-8): the normalisation in front of the SSIM (same notebook cell, `loss`)."""
    lo, hi = img.min(), img.max()
    return (img - lo) / (hi - lo + 1e-8)
