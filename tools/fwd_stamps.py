#!/usr/bin/env python3
"""Diagnostic: where a forward wave spends its life (build with -DDIFFUS_STAMP): per-phase cycles from s_memtime stamps
of lane 0 of every wave, plus when waves start (generations of the grid).  Shares, not absolute kernel time."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DIFFUS_LIB"] = os.path.abspath(sys.argv[1])
from diffus_amd import CapturedStep, _lib
from diffus_amd.phantom import phantom, pose_ring
lib = _lib.load()
P = int(os.environ.get("POSES", "32"))
vol = torch.from_numpy(phantom(256)).cuda()
src, dirs = pose_ring(256, P, 256)
hp = CapturedStep(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), 512, 1e-4, "trilinear")
nw = P * 256
st = torch.zeros(nw * 8, dtype=torch.int64, device="cuda")
lib.diffus_debug_set_fwd_stamps.argtypes = [C.c_void_p]
assert lib.diffus_debug_set_fwd_stamps(C.c_void_p(st.data_ptr())) == 0
for _ in range(3):
    st.zero_(); hp.fwd()
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(nw, 8)[:, :7].astype(np.int64)
t0 = s[:, 0].min()
names = ["pose+gather (addresses, loads, lerp)", "-> chunked (LDS)", "reflect", "scan + sweep", "attenuation", "-> interleaved (LDS)", ]
d = np.diff(s, axis=1)
life = s[:, 6] - s[:, 0]
print("waves %d   s_memtime ticks (100 MHz constant clock? or shader cycles -- use shares)" % nw)
print("wave lifetime: mean %.0f median %.0f max %.0f;  kernel span %.0f" % (life.mean(), np.median(life), life.max(), s[:, 6].max() - t0))
for i, n in enumerate(names):
    print("  %-40s mean %8.0f  median %8.0f  share %5.1f %%" % (n, d[:, i].mean(), np.median(d[:, i]), 100 * d[:, i].sum() / life.sum()))
st_rel = np.sort(s[:, 0] - t0)
for q in (0.1, 0.25, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0):
    print("  %3.0f %% of the waves have started by %8.0f (%.0f %% of the span)" % (100 * q, st_rel[int(q * nw) - 1], 100 * st_rel[int(q * nw) - 1] / (s[:, 6].max() - t0)))
