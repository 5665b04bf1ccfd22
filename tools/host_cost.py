#!/usr/bin/env python3
"""Diagnostic: host time of one eager step (P small: the GPU is faster than the host) and of its parts."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffus_amd import CapturedStep, _lib  # noqa: E402
from diffus_amd.captured import _stream_id  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402

N = 256
P = int(os.environ.get("POSES", "1")); R = int(os.environ.get("RAYS", "256")); S = int(os.environ.get("SAMPLES", "512"))
IT = 20000
vol = torch.from_numpy(phantom(N)).cuda()
src, dirs = pose_ring(N, P, R)
hp = CapturedStep(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), S, 1e-4, "trilinear", layout="paired")


def per_call(fn, n=IT):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


print("P=%d R=%d S=%d   (host us per call / wall us per call incl. the final drain)" % (P, R, S))
print("step()            %.2f / %.2f" % per_call(hp.step))
print("step_mse(ALL)     %.2f / %.2f" % per_call(lambda: hp.step_mse(_lib.BWD_ALL)))
print("finish_grad()     %.2f / %.2f" % per_call(hp.finish_grad))
print("_stream_id        %.2f / %.2f" % per_call(lambda: _stream_id(hp.dev)))
print("_inputs_now       %.2f / %.2f" % per_call(hp._inputs_now))
print("zero_grad         %.2f / %.2f" % per_call(hp.zero_grad))
print("empty lambda      %.2f / %.2f" % per_call(lambda: None))
