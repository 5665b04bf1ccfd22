#!/usr/bin/env python3
"""tests/test_tilted_fans.py::test_random_coplanar_fans_medium_size over many seeds (volumes of 100-200 voxels a side, 96 rays x 400 steps,
any orientation: patches that fill their tiles, several passes, row chunks).  python tools/fuzz_slab_medium.py first count"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import diffus_amd as da
import test_tilted_fans as t
bad = 0
t0 = time.time()
first, count = int(sys.argv[1]), int(sys.argv[2])
for k, seed in enumerate(range(first, first + count)):
    try:
        t.test_random_coplanar_fans_medium_size(da, seed)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, str(e)[:300], flush=True)
    if (k + 1) % 10 == 0:
        print(k + 1, "seeds,", bad, "failures, %.0f s" % (time.time() - t0), flush=True)
print("done:", count, "seeds from", first, ",", bad, "failures")
