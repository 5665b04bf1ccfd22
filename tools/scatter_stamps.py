#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of scatter_patch_kernel (build with -DDIFFUS_STAMP)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DIFFUS_LIB"] = os.path.abspath(sys.argv[1])
from diffus_amd import CapturedStep as HotPath
from diffus_amd import _lib
from diffus_amd.phantom import phantom, pose_ring
lib = _lib.load()
vol = torch.from_numpy(phantom(256)).cuda()
src, dirs = pose_ring(256, 32, 256)
hp = HotPath(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), 512, 1e-4, "trilinear", sparse=False)
hp.fwd(); hp.loss_and_grad(); hp.zero_grad(); hp.bwd(_lib.BWD_SCAN)
nblk = 32 * 16 * 8
nfin = 32 * 8  # the finishing row in front of the patch rows: P * ray groups blocks, the first P of them work
st = torch.zeros((nblk + nfin) * 8, dtype=torch.int64, device="cuda")
lib.diffus_debug_set_stamps.argtypes = [C.c_void_p]
assert lib.diffus_debug_set_stamps(C.c_void_p(st.data_ptr())) == 0
for _ in range(3):
    st.zero_(); hp.zero_grad(); hp.bwd(_lib.BWD_SCATTER)
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(nblk + nfin, 8)[nfin:]
done = s[:, 7] == 1
print("blocks", nblk, "completed-with-tile", done.sum(), "early-exit/fallback", (~done).sum())
d = np.diff(s[done][:, :6].astype(np.int64), axis=1)
names = ["load+cells", "bbox-reduce", "zero", "lds-add", "flush"]
tot = (s[done][:, 5] - s[done][:, 0])
print("cycles/block: mean %.0f median %.0f max %.0f" % (tot.mean(), np.median(tot), tot.max()))
for i, n in enumerate(names):
    print("  %-12s mean %8.0f  median %8.0f  max %8.0f  share %.1f%%" % (n, d[:, i].mean(), np.median(d[:, i]), d[:, i].max(), 100 * d[:, i].sum() / tot.sum()))
print("tile entries: mean %.0f max %.0f" % (s[done][:, 6].mean(), s[done][:, 6].max()))
nd = s[~done]
if len(nd):
    e = (nd[:, 2] > 0)
    print("not-done blocks that reached bbox:", e.sum())
span = s[:, :6][s[:, :6] > 0]
print("kernel span (cycles, min start..max end): %.0f" % (span.max() - span.min()))
te = s[done][:, 6]
for c in (1024, 2048, 3072, 4096, 6144, 8192, 12288):
    print("tile entries <= %5d: %5.1f %%" % (c, 100.0 * (te <= c).mean()))
# blocks by tile size: how much of the kernel's block time do the small (rays clamped outside the volume) patches take?
life = (s[:, 5] - s[:, 0]).astype(np.float64)
ok = (s[:, 5] > 0) & (s[:, 0] > 0)
te_all = s[:, 6].astype(np.float64)
for lo, hi in ((0, 1), (1, 64), (64, 256), (256, 1024), (1024, 2048), (2048, 4096)):
    m = ok & (te_all >= lo) & (te_all < hi)
    if m.any():
        print("tile entries [%4d, %4d): %5d blocks  mean life %7.0f cycles  share of block time %5.1f %%" % (lo, hi, m.sum(), life[m].mean(), 100 * life[m].sum() / life[ok].sum()))
print("blocks without stamps at the end (returned early):", int((~ok).sum()))
dd = np.diff(s[:, :6].astype(np.int64), axis=1)
for lo, hi in ((1, 64), (64, 256), (256, 1024), (1024, 4096)):
    m = ok & (te_all >= lo) & (te_all < hi)
    if m.any():
        print("tile [%4d, %4d): " % (lo, hi) + "  ".join("%s %6.0f" % (n, dd[m][:, i].mean()) for i, n in enumerate(names)))
