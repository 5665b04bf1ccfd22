#!/usr/bin/env python3
"""Diagnostic: one case of tools/fuzz_one_pass.py (seed, index) against the float64 oracle -- are the one-pass step and the
two-call path each within the frame's own conditioning, or is one of them wrong?   usage: diag_fuzz_case.py SEED INDEX ..."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import diffus_amd as da  # noqa: E402
from oracle.conditioning import frame64_and_tolerance  # noqa: E402
from tools.fuzz_one_pass import gen_case  # noqa: E402

seed = int(args[0])
for index in (int(a) for a in args[1:]):
    rng = np.random.default_rng(seed)
    for _ in range(index + 1):
        k = gen_case(rng)
    v = torch.from_numpy(k["vol"]).cuda()
    s, d = torch.from_numpy(k["src"]).cuda(), torch.from_numpy(k["dirs"]).cuda()
    t = torch.from_numpy(k["tgt"]).cuda()
    one = da.CapturedStep(v, s, d, k["S"], k["alpha"], k["sampler"], start=k["start"], layout=k["layout"], target=t, loss_scale=k["scale"])
    one.step()
    f2 = da.render_poses(v, s, d, k["S"], k["alpha"], start=k["start"], sampler=k["sampler"], layout=k["layout"])
    torch.cuda.synchronize()
    print("case", seed, index, k["dims"], "P", k["P"], "R", k["R"], "S", k["S"], "start", k["start"], k["sampler"], k["layout"], k["alpha"], k["src"].dtype)
    for p in range(k["P"]):
        f64, tol, sens = frame64_and_tolerance(k["vol"], k["src"][p], k["dirs"][p], k["S"], k["alpha"], sampler=k["sampler"])
        den = np.abs(f64).max()
        e1 = np.abs(one.frame[p].cpu().numpy() - f64).max() / den
        e2 = np.abs(f2[p].cpu().numpy() - f64).max() / den
        e12 = float((one.frame[p] - f2[p]).abs().max()) / den
        print("   pose %d: max|frame| %.3g  one-pass vs f64 %.2e  two-call vs f64 %.2e  between them %.2e  sens %.2e  tol %.2e" % (p, den, e1, e2, e12, sens, tol))
