#!/bin/bash
# GPU box: kernel trace of the drop-in demo frame (tools/demo_frame_loop.py): per-kernel median duration and gap to the next.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_demo1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_demo1 -- python3 tools/demo_frame_loop.py > gpurun_out/demo1.log 2>&1
tail -1 gpurun_out/demo1.log
python3 - <<PY
import csv,glob,collections,statistics
f=glob.glob("gpurun_out/prof_demo1/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
rows=rows[len(rows)//2:]
names=[r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("<")[0].split("(")[0] for r in rows]
idx=[i for i,n in enumerate(names) if n=="median_kernel"]
per=idx[1]-idx[0]
rows=rows[idx[0]:]; names=names[idx[0]:]
print("kernels per frame",per)
dur=collections.defaultdict(list); gap=collections.defaultdict(list)
for k in range(0,len(rows)-per-1,per):
    for j in range(per):
        r=rows[k+j]; nx=rows[k+j+1]
        dur[(j,names[k+j])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
        gap[(j,names[k+j])].append((int(nx["Start_Timestamp"])-int(r["End_Timestamp"]))/1e3)
td=tg=0
for key in sorted(dur):
    d=statistics.median(dur[key]); g=statistics.median(gap[key]); td+=d; tg+=g
    print("%2d %-40s dur %6.2f us  gap to next %6.2f us"%(key[0],key[1][:40],d,g))
print("sum of durations %.1f us, of gaps %.1f us"%(td,tg))
PY
