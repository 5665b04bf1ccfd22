#!/usr/bin/env python3
"""The call sequence of the reference's `[DEMO] REUBEN DATA 46` notebook (cells 11-14) with
`from src.renderer import *` swapped for `from diffus_amd import *`, on the analytic phantom
(the ReMIND volumes are not shipped with the reference):

    generate_cone_directions -> UltrasoundRenderer(...).plot_beam_frame(artifacts=True, start=...)
    -> rotate_around_apex -> differentiable_splat -> image

    python examples/reuben_like_demo.py [out.png]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffus_amd import *  # noqa: F401,F403,E402
from diffus_amd.phantom import phantom  # noqa: E402


def run(n=256, n_rays=256, d1=52, d2=185, seed=0):
    Z_vol = torch.from_numpy(phantom(n)).cuda()
    source = torch.tensor([88.0769, -11.5385, 110.0], dtype=torch.float64)     # an apex outside the volume, f64
    direction_mri_world = np.array([0.35, 0.94])
    opening_angle = np.radians(52.47)
    directions = generate_cone_directions(direction_mri_world, opening_angle, n_rays)
    renderer = UltrasoundRenderer(num_samples=d2, attenuation_coeff=1e-4)
    x, y, z, intensities = renderer.plot_beam_frame(volume=Z_vol, source=source, directions=directions,
                                                    angle=np.degrees(opening_angle) / 2 - 5, plot=False, artifacts=True,
                                                    start=d1 - 12, seed=seed)
    xr, yr = rotate_around_apex(x.flatten().float(), y.flatten().float(), (128.0, 10.0), tuple(direction_mri_world))
    img = differentiable_splat(xr.reshape(x.shape), yr.reshape(y.shape), z, intensities, H=n, W=n, sigma=1)
    return x, y, z, intensities, img


if __name__ == "__main__":
    x, y, z, I, img = run()
    print("frame", tuple(I.shape), I.dtype, "image", tuple(img.shape), img.dtype,
          "intensity range %.4f..%.4f" % (float(I.min()), float(I.max())))
    if len(sys.argv) > 1:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.imsave(sys.argv[1], img.cpu().numpy().T, cmap="gray", origin="lower")
        print("wrote", sys.argv[1])
