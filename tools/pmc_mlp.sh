#!/bin/bash
# GPU box: SQ counters for the fused MLP kernels (diagnostic).  Output: table on stdout.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_mlp; mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/time_mlp.py > $OUT/b$i.json 2> $OUT/e$i.err
  echo "pass $i rc=$?"
done
python3 - <<'PY'
import csv,glob,collections
res=collections.defaultdict(dict)
for f in glob.glob("gpurun_out/prof_mlp/p*/*/*counter_collection.csv"):
    tmp=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("<")[0].split("(")[0]
        tmp[(n,r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (n,c),v in tmp.items():
        v=sorted(v); res[n][c]=v[len(v)//2]
for n in ("mlp_fwd_kernel","mlp_bwd_kernel"):
    print("==",n)
    for c,v in sorted(res[n].items()): print("   %-30s %16.0f"%(c,v))
PY
