#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / median / max duration (us)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    by[name[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-72s n=%4d med=%8.1f max=%8.1f total=%9.1f us" % (k, len(v), v2[len(v2) // 2], v2[-1], sum(v)))
