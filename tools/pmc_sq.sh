#!/bin/bash
# GPU box: SQ instruction / stall counters for the hot kernels (diagnostic).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_sq; mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --eager > $OUT/b$i.json 2> $OUT/e$i.err
  echo "pass $i rc=$?"
done
python3 - <<'PY'
import csv,glob,collections
res=collections.defaultdict(dict)
for f in glob.glob("gpurun_out/prof_sq/p*/*/*counter_collection.csv"):
    rows=list(csv.DictReader(open(f)))
    tmp=collections.defaultdict(list)
    for r in rows:
        n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("<")[0].split("(")[0]
        tmp[(n,r["Counter_Name"])].append((int(r["Grid_Size"]),float(r["Counter_Value"])))
    for (n,c),v in tmp.items():
        g=max(x for x,_ in v); vals=sorted(t for x,t in v if x==g); res[n][c]=vals[len(vals)//2]
for n in ("render_fwd_kernel","render_bwd_kernel","scatter_patch_kernel"):
    print("==",n)
    for c,v in sorted(res[n].items()): print("   %-26s %16.0f"%(c,v))
PY
