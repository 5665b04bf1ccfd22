#!/usr/bin/env python3
"""Where the HOST time of one drop-in training step goes (render_poses -> loss -> backward): each piece called in a loop
on its own, enqueue time only (the device is drained outside the timed loops)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import diffus_amd as da  # noqa: E402
from diffus_amd import _lib, renderer as R  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402

P = int(os.environ.get("POSES", "32"))
vol = torch.from_numpy(phantom(256)).cuda().requires_grad_(True)
s, d = pose_ring(256, 32, 256)
s = torch.from_numpy(s[:P]).cuda().requires_grad_(True)
d = torch.from_numpy(d[:P]).cuda().requires_grad_(True)
lib = _lib.load()


def t(fn, n=300, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt


def full():
    f = da.render_poses(vol, s, d, 512, 1e-4, sampler="trilinear")
    (f * f).sum().backward()
    vol.grad = None; s.grad = None; d.grad = None


pb = R._Problem(vol, s, d, 512, 0, 1e-4, "trilinear", "auto")
frame = torch.empty((pb.P, pb.R, pb.N1), device="cuda")
ws = pb.workspace()
gb = R._gradbuf(pb.dev, pb.shape)
gsrc = torch.empty((P, 3), device="cuda"); gd = torch.empty((P, 256, 3), device="cuda")
dense = torch.empty(pb.shape, device="cuda")
vd, sd_, dd_ = vol.detach(), s.detach(), d.detach()    # (the same objects every call: the converted-volume cache is keyed by identity)
rows = [
    ("_Problem()", lambda: R._Problem(vol, s, d, 512, 0, 1e-4, "trilinear", "auto")),
    ("torch.empty(frame)", lambda: torch.empty((pb.P, pb.R, pb.N1), dtype=torch.float32, device=pb.dev)),
    ("pb.workspace()", pb.workspace),
    ("ctypes diffus_render_fwd", lambda: lib.diffus_render_fwd(*pb.common(), R._ptr(frame), None, R._ptr(ws), ws.numel(), R._stream(pb.dev))),
    ("ctypes diffus_render_bwd", lambda: lib.diffus_render_bwd(*pb.common(), R._ptr(frame), R._ptr(gb[0]), R._ptr(gb[1]), R._ptr(gsrc), R._ptr(gd), 3, R._ptr(ws), ws.numel(), R._stream(pb.dev))),
    ("ctypes gradbuf_flush DENSE", lambda: lib.diffus_gradbuf_flush(R._ptr(gb[0]), R._ptr(gb[1]), *pb.shape, R._ptr(dense), 3, R._stream(pb.dev))),
    ("render_poses no_grad", lambda: da.render_poses(vd, sd_, dd_, 512, 1e-4, sampler="trilinear")),
    ("render_poses (autograd node)", lambda: da.render_poses(vol, s, d, 512, 1e-4, sampler="trilinear")),
]
for name, fn in rows:
    print("%-32s %7.1f us" % (name, t(fn)))
f = da.render_poses(vol, s, d, 512, 1e-4, sampler="trilinear")
print("%-32s %7.1f us" % ("(f*f).sum()", t(lambda: (f * f).sum())))


def bwd_only():
    ff = da.render_poses(vol, s, d, 512, 1e-4, sampler="trilinear")
    l = (ff * ff).sum()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    l.backward()
    dt = time.perf_counter() - t0
    vol.grad = None; s.grad = None; d.grad = None
    return dt


for _ in range(10):
    bwd_only()
print("%-32s %7.1f us" % ("loss.backward() call (host)", sum(bwd_only() for _ in range(100)) / 100 * 1e6))
print("%-32s %7.1f us" % ("whole step (wall, drained)", t(full, 100) ))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    full()
torch.cuda.synchronize()
print("%-32s %7.1f us" % ("whole step (wall incl. device)", (time.perf_counter() - t0) / 200 * 1e6))
# device time of the step's kernels alone
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50):
    lib.diffus_render_fwd(*pb.common(), R._ptr(frame), None, R._ptr(ws), ws.numel(), R._stream(pb.dev))
    lib.diffus_render_bwd(*pb.common(), R._ptr(frame), R._ptr(gb[0]), R._ptr(gb[1]), R._ptr(gsrc), R._ptr(gd), 3, R._ptr(ws), ws.numel(), R._stream(pb.dev))
    lib.diffus_gradbuf_flush(R._ptr(gb[0]), R._ptr(gb[1]), *pb.shape, R._ptr(dense), 3, R._stream(pb.dev))
e1.record(); torch.cuda.synchronize()
print("%-32s %7.1f us" % ("device: fwd + bwd + dense flush", e0.elapsed_time(e1) / 50 * 1e3))
# the same step with the autograd engine kept on the calling thread (no hand-off to the device worker thread)
with torch.autograd.set_multithreading_enabled(False):
    for _ in range(10):
        bwd_only()
    print("%-32s %7.1f us" % ("backward(), single-threaded engine", sum(bwd_only() for _ in range(100)) / 100 * 1e6))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        full()
    torch.cuda.synchronize()
    print("%-32s %7.1f us" % ("whole step, single-threaded", (time.perf_counter() - t0) / 200 * 1e6))
import cProfile, pstats
with torch.autograd.set_multithreading_enabled(False):
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200):
        full()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
