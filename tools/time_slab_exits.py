#!/usr/bin/env python3
"""Diagnostic: the scatter kernel on a tilted fan (ROLL / PITCH, default 20 / 0) for several library builds -- the stage-exit
probes of the slab path (-DDIFFUS_SLAB_EXIT=n).  Usage: tools/time_slab_exits.py lib1.so lib2.so ...  (each in its own process)"""
import os
import subprocess
import sys

if len(sys.argv) > 2 or (len(sys.argv) == 2 and not sys.argv[1].endswith(".so")):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, lib])
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DIFFUS_LIB"] = os.path.abspath(sys.argv[1])
import torch  # noqa: E402

from bench import time_events  # noqa: E402
from diffus_amd import CapturedStep, _lib  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402

roll, pitch = float(os.environ.get("ROLL", "20")), float(os.environ.get("PITCH", "0"))
vol = torch.from_numpy(phantom(256)).cuda()
src, dirs = pose_ring(256, 32, 256, roll_deg=roll, pitch_deg=pitch)
hp = CapturedStep(vol, torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda(), 512, 1e-4, "trilinear", fans=os.environ.get("FANS", "auto"))
for _ in range(5):
    hp.step()
scat = time_events(lambda: hp.bwd(_lib.BWD_SCATTER), 200, pre=hp.finish_grad)
print("%-40s roll %g pitch %g: scatter %.2f us median, %.2f min" % (os.path.basename(sys.argv[1]), roll, pitch, scat["median"] * 1e3, scat["min"] * 1e3), flush=True)
