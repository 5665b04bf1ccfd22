#!/usr/bin/env python3
"""Diagnostic: where an adjoint-scan wave spends its life (build with -DDIFFUS_STAMP): per-phase cycles from s_memtime
stamps of lane 0 of every wave.  Shares, not absolute kernel time (the stamps and their fences cost cycles themselves)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DIFFUS_LIB"] = os.path.abspath(sys.argv[1])
from diffus_amd import CapturedStep, _lib
from diffus_amd.phantom import phantom, pose_ring
lib = _lib.load()
P = int(os.environ.get("POSES", "32")); N = int(os.environ.get("N", "256")); R = int(os.environ.get("RAYS", "256")); S = int(os.environ.get("SAMPLES", "512"))
vol = torch.from_numpy(phantom(N)).cuda()
src, dirs = pose_ring(N, P, R)
hp = CapturedStep(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), S, 1e-4, "trilinear", fused_loss=False)
nw = P * R
st = torch.zeros(nw * 2 * 12, dtype=torch.int64, device="cuda")
lib.diffus_debug_set_bwd_stamps.argtypes = [C.c_void_p]
assert lib.diffus_debug_set_bwd_stamps(C.c_void_p(st.data_ptr())) == 0
hp.fwd(); hp.loss_and_grad()
for _ in range(3):
    st.zero_(); hp.bwd(_lib.BWD_SCAN)
torch.cuda.synchronize()
full = st.cpu().numpy().reshape(nw * 2, 12).astype(np.int64)
full = full[full[:, 0] > 0]
s = full[:, :10]
names = ["pose + gather + stash + -> chunked", "gframe load + reflect", "local product + forward scan", "P' / seeds loop",
         "sweep from 0 (A part)", "reverse scan", "sweep with U_in + rbar", "zbar + store", "pose gradient"]
d = np.diff(s, axis=1)
life = s[:, 9] - s[:, 0]
print("waves %d (P=%d R=%d S=%d); wave lifetime: mean %.0f median %.0f max %.0f cycles" % (len(s), P, R, S, life.mean(), np.median(life), life.max()))
for i, n in enumerate(names):
    print("  %-38s mean %8.0f  median %8.0f  share %5.1f %%" % (n, d[:, i].mean(), np.median(d[:, i]), 100 * d[:, i].sum() / life.sum()))
rt0, rt1 = full[:, 10], full[:, 11]
t0 = rt0.min()
span = (rt1.max() - t0) / 100.0
print("wall clock (s_memrealtime, 100 MHz): first wave start -> last wave end %.1f us; a wave lives %.1f us on average (=> %.2f GHz shader clock)"
      % (span, (rt1 - rt0).mean() / 100.0, life.mean() / ((rt1 - rt0).mean() / 100.0) / 1e3))
st_us = np.sort(rt0 - t0) / 100.0
en_us = np.sort(rt1 - t0) / 100.0
for q in (0.05, 0.1, 0.2, 0.25, 0.3, 0.4, 0.45, 0.5, 0.55, 0.6, 0.7, 0.75, 0.8, 0.9, 0.95, 1.0):
    i = int(q * len(st_us)) - 1
    print("  %3.0f %% of the waves have started by %5.1f us, ended by %5.1f us" % (100 * q, st_us[i], en_us[i]))

# how many waves are resident over time (per SIMD: / 1024)
ts = np.linspace(0, span, 35)
res = [(int(((rt0 - t0) / 100.0 <= t).sum() - ((rt1 - t0) / 100.0 <= t).sum())) for t in ts]
print("resident waves per SIMD over time (us: waves):", "  ".join("%.0f:%.1f" % (t, r / 1024.0) for t, r in zip(ts, res)))
