#!/bin/bash
# GPU box: kernel trace of examples/train_ssim_chain.py; prints the kernels of one captured iteration (the last complete
# period of the trace) with their durations and the gap to the next kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_chain
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_chain -- python3 examples/train_ssim_chain.py > gpurun_out/chain.log 2>&1
tail -1 gpurun_out/chain.log
python3 - <<PY
import csv,glob,collections,statistics
f=glob.glob("gpurun_out/prof_chain/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("<")[0].split("(")[0] for r in rows]
# period = distance between the last occurrences of the adjoint-scan kernel
idx=[i for i,n in enumerate(names) if n=="render_bwd_kernel"]
per=idx[-1]-idx[-2]
k0=idx[-10]
print("kernels per iteration",per)
dur=collections.defaultdict(list); gap=collections.defaultdict(list)
for k in range(k0,idx[-2],per):
    for j in range(per):
        r=rows[k+j]; nx=rows[k+j+1]
        dur[(j,names[k+j])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
        gap[(j,names[k+j])].append((int(nx["Start_Timestamp"])-int(r["End_Timestamp"]))/1e3)
td=tg=0
for key in sorted(dur):
    d=statistics.median(dur[key]); g=statistics.median(gap[key]); td+=d; tg+=g
    print("%2d %-44s dur %6.2f us  gap %6.2f us"%(key[0],key[1][:44],d,g))
print("sum of durations %.1f us, of gaps %.1f us"%(td,tg))
PY
