#!/usr/bin/env python3
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-600:]          # the steady state: the last 200 replays
gaps = collections.defaultdict(list)
dur = collections.defaultdict(list)
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]
for a, b in zip(rows, rows[1:]):
    gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for r in rows:
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in dur.items():
    v.sort(); print("kernel %-26s n=%4d median %.2f us" % (k, len(v), v[len(v) // 2]))
for k, v in gaps.items():
    v.sort(); print("gap %-26s -> %-26s n=%4d median %.2f us" % (k[0], k[1], len(v), v[len(v) // 2]))
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("per step (3 kernels): %.2f us" % (span / (len(rows) / 3)))
