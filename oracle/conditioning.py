"""How close can ANY float32 evaluation of a frame be to the float64 one?  TEST INFRASTRUCTURE ONLY (like the rest of
oracle/): used by tests/ and by bench.py's after-the-run self-check, never by diffus_amd/.

Rays that graze the skull leave bone for air through a few samples of falling impedance; the echo there is b/d with d
nearly cancelled (|echo| reaches 100-300 on the benchmark phantom) and every float32 evaluation -- the reference's dense
LU included, golden G17: 3.3e-5 and 1.4e-4 from its own float64 result on two such rays of BASELINE config 3 -- carries
(condition number) x eps of noise.  The parity tolerance on such a frame is therefore derived from the frame itself:

    sens = max-norm-relative change of the float64 frame when every impedance sample is rounded once more
           (relative +-2^-24, random signs)
    tol  = max(base, roundings x sens)

i.e. the usual `base` (2e-5) wherever the frame is well conditioned, and "10 input roundings' worth" where it is not (a
512-step scan accumulates ~sqrt(512) = 22 roundings; random signs under-estimate the worst case by a few times:
measured on MI355X at config 3, the kernels sit at 0.3-6 x sens on every pose, tools/diag_conditioning.py).
The factor is pinned on the reference (round 4, golden G19 = every ray of poses 18 and 6 through the reference's dense
solves in float32, in float64, and as a float64 pipeline): the reference's own float32 frame is 4.2e-5 / 1.7e-5 from
the float64 pipeline on those poses, and tests/test_ill_conditioned.py holds this tolerance between 1x and 4x of
max(1e-5, 2 x that) -- 3.0e-4 on pose 18 (3.6x), 4.9e-5 on pose 6 (1.5x).  (Rounds 2-3 used 16 roundings: 4.9e-4 on pose
18, 5.7x the reference-derived bound.)
Restates reference src/renderer.py:33 (reflection), :412-457 (echo series, via the O(N) form pinned by G1-G4/G17) and
:256-259 (attenuation) in float64.
"""
from __future__ import annotations

import numpy as np

from . import oracle as orc


def frame64_and_tolerance(vol, source, directions, S, alpha, sampler="trilinear", base=2e-5, roundings=10, trials=16, seed=0):
    """start = 0 only.  -> (frame64 (R,S) float64, tol, sens).  The samples are the float32 ones of the reference's
    march (orc.sample_*), everything after them is float64."""
    if sampler == "trilinear":
        imp = orc.sample_trilinear(vol, source, directions, S)
    else:
        imp = orc.sample_nearest(vol, source, directions, S)[3]
    z = imp.astype(np.float64)
    att = np.exp(-float(alpha) * np.arange(S, dtype=np.float64))

    def frame_of(zz):
        with np.errstate(divide="ignore", invalid="ignore"):
            r = (zz[:, 1:] - zz[:, :-1]) / (zz[:, :-1] + zz[:, 1:])
        return orc.echo_scan(r, np.float64) * att

    # r is formed in float32 by the reference (and by the kernels): frame64 starts from THAT r
    f64 = orc.echo_scan(orc.reflection(imp).astype(np.float64), np.float64) * att
    ref = frame_of(z)
    den = max(float(np.abs(ref).max()), 1e-300)
    rng = np.random.default_rng(seed)
    sens = 0.0
    for _ in range(trials):
        pert = z * (1.0 + rng.choice([-1.0, 1.0], size=z.shape) * 2.0 ** -24)
        sens = max(sens, float(np.abs(frame_of(pert) - ref).max()) / den)
    return f64, max(base, roundings * sens), sens
