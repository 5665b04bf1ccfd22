#!/bin/bash
# GPU box: texture-addresser / L1 (TCP) counters for the gather kernels (diagnostic): is the per-CU memory pipe the limiter?
# One or two counters per pass (the TA/TCP blocks have few slots; an over-subscribed pass aborted the process), each pass
# under its own timeout, progress appended to gpurun_out/prof_ta/progress.log.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_ta; mkdir -p $OUT
i=0
for C in "TA_TA_BUSY_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-callers --eager $* > $OUT/b$i.json 2> $OUT/e$i.err
  rc=$?
  echo "pass $i rc=$rc ($C)" | tee -a $OUT/progress.log
  if [ $rc -ge 124 ]; then echo "timed out: stopping" | tee -a $OUT/progress.log; break; fi
done
python3 - <<'PY'
import csv,glob,collections
res=collections.defaultdict(dict)
for f in glob.glob("gpurun_out/prof_ta/p*/**/*counter_collection.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    tmp=collections.defaultdict(list)
    for r in rows:
        n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("<")[0].split("(")[0]
        tmp[(n,r["Counter_Name"])].append((int(r["Grid_Size"]),float(r["Counter_Value"])))
    for (n,c),v in tmp.items():
        g=max(x for x,_ in v); vals=sorted(t for x,t in v if x==g); res[n][c]=vals[len(vals)//2]
for n in ("render_fwd_kernel","render_bwd_kernel","scatter_patch_kernel"):
    print("==",n)
    for c,v in sorted(res[n].items()): print("   %-40s %16.0f"%(c,v))
PY
