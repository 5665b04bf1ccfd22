import sys, os, torch, ctypes as C
sys.path.insert(0, os.getcwd())
from diffus_amd import _lib
from bench import time_events
lib = _lib.load()
for n in (256, 512):
    v = torch.randn(n, n, n, device="cuda")
    out = torch.empty(lib.diffus_paired_floats(n, n, n), device="cuda")
    f = lambda: lib.diffus_pair_volume(C.c_void_p(v.data_ptr()), n, n, n, C.c_void_p(out.data_ptr()), None)
    f(); torch.cuda.synchronize()
    t = time_events(f, 20)
    gb = (v.numel() * 4 + out.numel() * 4) / 1e9
    print(n, "pair_volume median %.1f us  %.0f GB/s (read + write, no re-reads)" % (t["median"] * 1e3, gb / (t["median"] * 1e-3)))
