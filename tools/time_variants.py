#!/usr/bin/env python3
"""Diagnostic: event-time the hot kernels for several library builds, one child process per build (a process binds
one libdiffus_hip.so).  usage: tools/time_variants.py lib1.so lib2.so ...   [env POSES, N, RAYS, SAMPLES]"""
import os, subprocess, sys
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_scatter.py"), lib],
                       capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
