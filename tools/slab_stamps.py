#!/usr/bin/env python3
"""Diagnostic: where a block of the scatter kernel's SLAB path (fans that leave the slice) spends its cycles.
Build: python -m diffus_amd.build -DDIFFUS_STAMP -o/path/lib.so ; run: tools/slab_stamps.py lib.so [roll pitch]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DIFFUS_LIB"] = os.path.abspath(sys.argv[1])
from diffus_amd import CapturedStep, _lib
from diffus_amd.phantom import phantom, pose_ring
roll = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
pitch = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
lib = _lib.load()
vol = torch.from_numpy(phantom(256)).cuda()
src, dirs = pose_ring(256, 32, 256, roll_deg=roll, pitch_deg=pitch)
hp = CapturedStep(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), 512, 1e-4, "trilinear", sparse=False)
hp.fwd(); hp.loss_and_grad(); hp.zero_grad(); hp.bwd(_lib.BWD_SCAN)
nblk = 32 * 16 * 8
nfin = 32 * 8
st = torch.zeros((nblk + nfin) * 8, dtype=torch.int64, device="cuda")
lib.diffus_debug_set_stamps.argtypes = [C.c_void_p]
assert lib.diffus_debug_set_stamps(C.c_void_p(st.data_ptr())) == 0
for _ in range(3):
    st.zero_(); hp.zero_grad(); hp.bwd(_lib.BWD_SCATTER)
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(nblk + nfin, 8)[nfin:].astype(np.float64)
ok = (s[:, 5] > 0) & (s[:, 0] > 0)
print("roll %.0f pitch %.0f: blocks %d, through the slab path %d" % (roll, pitch, nblk, ok.sum()))
s = s[ok]
life = s[:, 5] - s[:, 0]
print("cycles per block: mean %.0f median %.0f max %.0f" % (life.mean(), np.median(life), life.max()))
first_add = np.mod(s[:, 7], 2 ** 24); s[:, 7] = np.floor(s[:, 7] / 2 ** 24)
chunks = np.mod(s[:, 7], 16); entries = np.floor(s[:, 7] / 16)
s[:, 7] = entries
parts = {"pose, loads, tile clear": s[:, 1] - s[:, 0], "plane+cells+classes": s[:, 2] - s[:, 1], "records+barrier": s[:, 6] - s[:, 2],
         "adds (+barrier)": s[:, 3], "flush (+barrier)": s[:, 4], "set-up of the first pass": first_add, "rest": life - (s[:, 6] - s[:, 0]) - s[:, 3] - s[:, 4] - first_add}
s[:, 6] = chunks
for k, v in parts.items():
    print("  %-18s mean %7.0f  median %7.0f  share %5.1f %%" % (k, v.mean(), np.median(v), 100 * v.sum() / life.sum()))
print("pass-chunks per block: mean %.2f  max %d;  histogram %s" % (s[:, 6].mean(), s[:, 6].max(), np.bincount(s[:, 6].astype(int))[:10]))
print("tile entries per block (columns x slots, all chunks): mean %.0f median %.0f max %.0f" % (s[:, 7].mean(), np.median(s[:, 7]), s[:, 7].max()))
span = np.concatenate([s[:, 0], s[:, 5]])
print("kernel span (cycles): %.0f" % (span.max() - span.min()))
for lo, hi in ((1, 2), (2, 3), (3, 5), (5, 100)):
    m = (s[:, 6] >= lo) & (s[:, 6] < hi)
    if m.any():
        print("pass-chunks [%d, %d): %5d blocks  life %7.0f  adds %6.0f  flush %6.0f  entries %6.0f" % (lo, hi, m.sum(), life[m].mean(), s[m, 3].mean(), s[m, 4].mean(), s[m, 7].mean()))
