#!/usr/bin/env python3
"""Fuzz compute_echo_traces (diffus_echo_traces; reference src/renderer.py:439-457) against the C oracle's float64 series: random row
counts and lengths (0 ... 3000: one launch, and rows walked in 1024-sample pieces), coefficient scales from tissue-like to |r| -> 1,
NaN and zero rows.  The bar is tests/test_hip_parity.py::test_echo_traces_sizes': a row may be as far from float64 as 10x (30x in
pieces) the sequential float32 oracle's own error, floor 2e-5; NaN placement and the leading zero exact.

    python tools/fuzz_echo.py [first_seed] [count]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffus_amd as da  # noqa: E402
from oracle import oracle as orc  # noqa: E402

def gen(seed):
    """-> r (B, N) float32 of case `seed`."""
    rng = np.random.default_rng(90000 + seed)
    B = int(rng.integers(1, 7))
    N = int(rng.choice([0, 1, 2, 5, 63, 64, 65, 127, 128, 200, 256, 300, 511, 512, 513, 700, 1023, 1024, 1025, 1500, 2048, 3000]))
    a = float(rng.choice([0.02, 0.1, 0.3, 0.6, 0.9]))
    r = rng.uniform(-a, a, size=(B, N)).astype(np.float32)
    if N > 8:
        if rng.random() < 0.3:
            r[rng.integers(0, B), rng.integers(0, N)] = float(rng.choice([0.99995, -0.99995, 1.0, -1.0]))
        if rng.random() < 0.2:
            r[rng.integers(0, B), rng.integers(0, N)] = np.nan
        if rng.random() < 0.2:
            r[rng.integers(0, B), :] = 0
        if rng.random() < 0.2:      # alternating strong reflectors: |P| doubles every two steps (the renormalisation path)
            i = rng.integers(0, B)
            r[i] = np.where(np.arange(N) % 2 == 0, 0.9995, -0.9995)
    return r, a


def main():
    orc.build()
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    bad, worst_ratio = 0, 0.0
    t0 = time.time()
    for k, seed in enumerate(range(first, first + count)):
        r, a = gen(seed)
        B, N = r.shape
        e = da.compute_echo_traces(torch.from_numpy(r).cuda())[0].cpu().numpy()
        with np.errstate(all="ignore"):
            ref = orc.echo_scan(r.astype(np.float64), np.float64)
            o32 = orc.echo_scan(r, np.float32)
        why = None
        if e.shape != (B, N + 1) or not np.all(e[:, 0] == 0):
            why = "shape / leading zero"
        elif not np.all(np.isfinite(e)):
            why = "non-finite output"          # (the reference's nan_to_num: past a NaN coefficient every echo is 0)
        else:
            for i in range(B):
                den = float(np.max(np.abs(ref[i]))) or 1.0
                err = float(np.max(np.abs(e[i] - ref[i])) / den)
                noise = float(np.max(np.abs(o32[i] - ref[i])) / den)
                tol = max(2e-5, (10 if N < 1024 else 30) * noise)
                if noise > 0:
                    worst_ratio = max(worst_ratio, err / max(noise, 2e-6))
                if not err < tol:
                    why = "row %d: %.2e from float64, float32 oracle %.2e, max |echo| %.1f" % (i, err, noise, float(np.max(np.abs(ref[i]))))
                    break
        if why:
            bad += 1
            print("FAIL seed %d B %d N %d scale %.2f: %s" % (seed, B, N, a, why), flush=True)
        if (k + 1) % 500 == 0:
            print("%d cases, %d failures, worst error / float32 oracle's error %.1f, %.0f s" % (k + 1, bad, worst_ratio, time.time() - t0), flush=True)
    print("done: %d cases from %d, %d failures" % (count, first, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
