"""Host-side mirror of `rasterize_fan` (reference src/renderer.py:626-653), the one name of `from src.renderer import *`
that `[DEMO] REUBEN DATA 46` (cell 11) uses beside the hot path.  It is a call into SciPy's Delaunay-based
`scipy.interpolate.griddata` (third-party; `scipy>=1.7` in the reference's requirements.txt), not GPU work: the scattered
(x, z, intensity) samples are interpolated linearly back onto the grid spanned by THEIR OWN coordinates -- an
(n, n) array for n samples, rows following z, columns following x; `output_shape` is accepted and, as in the reference,
not used.  Points outside the samples' convex hull get 0.  Scan conversion on the GPU is `differentiable_splat`."""
from __future__ import annotations

import numpy as np


def _host_array(a) -> np.ndarray:
    if hasattr(a, "detach"):            # a torch tensor, possibly on the GPU
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def rasterize_fan(x_coords, z_coords, intensities, output_shape=(256, 256)) -> np.ndarray:
    from scipy.interpolate import griddata
    x, z, v = _host_array(x_coords), _host_array(z_coords), _host_array(intensities)
    samples = np.column_stack((x, z))                      # (n, 2) scattered sample positions
    cols, rows = np.meshgrid(x, z)                         # query grid: [i, j] = (x[j], z[i])
    return griddata(samples, v, (cols, rows), method="linear", fill_value=0)
