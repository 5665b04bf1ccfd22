#!/usr/bin/env python3
"""Times the fused impedance-MLP kernels on a 256^3 volume against the f32 matrix-core peak,
with the same network in plain torch layers beside it.  usage: python tools/time_mlp.py [n]"""
import json
import sys

import torch

sys.path.insert(0, ".")
import diffus_amd as da
from diffus_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = n ** 3
lib = _lib.load()
torch.manual_seed(0)
m = da.ImpedanceEstimator(1).cuda()
x = torch.randn(N, device="cuda")
params = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
y = torch.empty_like(x); gy = torch.randn_like(x); gx = torch.empty_like(x); gp = torch.empty(1153, device="cuda")
ws = torch.empty(lib.diffus_mlp_workspace_bytes(), dtype=torch.uint8, device="cuda")
mask = (torch.rand(N, device="cuda") < 0.35).to(torch.uint8)          # a head fills about a third of its box


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def fwd(msk=None):
    lib.diffus_mlp_fwd(x.data_ptr(), msk.data_ptr() if msk is not None else None, N, params.data_ptr(), 0.0, 1.0, 1.0, 400.0,
                       y.data_ptr(), 1, None)


def bwd(want_gx=True):
    lib.diffus_mlp_bwd(x.data_ptr(), None, N, params.data_ptr(), 0.0, 1.0, 1.0, gy.data_ptr(), 1, gp.data_ptr(),
                       gx.data_ptr() if want_gx else None, ws.data_ptr(), ws.numel(), None)


FLOP_FWD = 2 * (32 + 1024 + 32)          # per voxel
MFMA_FWD, MFMA_BWD = 2 * 1024, 4 * 2 * 1024   # flops issued on the matrix cores per voxel
out = {"voxels": N}
t = timeit(fwd); out["fwd_ms"] = t; out["fwd_TFLOPs_mfma"] = N * MFMA_FWD / t / 1e9
t = timeit(lambda: fwd(mask)); out["fwd_masked35_ms"] = t
t = timeit(bwd); out["bwd_ms"] = t; out["bwd_TFLOPs_mfma"] = N * MFMA_BWD / t / 1e9
t = timeit(lambda: bwd(False)); out["bwd_nogx_ms"] = t
out["peak_TFLOPs_f32_mfma"] = 157.3
out["fwd_frac"] = out["fwd_TFLOPs_mfma"] / 157.3; out["bwd_frac"] = out["bwd_TFLOPs_mfma"] / 157.3
# the same network as torch layers (rocBLAS GEMMs + elementwise kernels), chunked so the activations fit
xs = x.reshape(-1, 1)


def torch_fwd():
    with torch.no_grad():
        for c in xs.split(1 << 22):
            m.model(c)


def torch_fwd_bwd():
    for c, g in zip(xs.split(1 << 22), gy.reshape(-1, 1).split(1 << 22)):
        cr = c.clone().requires_grad_(True)
        (m.model(cr) * g).sum().backward()


out["torch_fwd_ms"] = timeit(torch_fwd, 3)
out["torch_fwd_bwd_ms"] = timeit(torch_fwd_bwd, 3)
out["fused_fwd_bwd_ms"] = out["fwd_ms"] + out["bwd_ms"]
print(json.dumps(out))
