#!/bin/bash
# GPU box: SQ instruction counters + kernel durations of the hot kernels for one library build (diagnostic).
# Usage: tools/pmc_step.sh TAG [library.so]      (env POSES/N/RAYS/SAMPLES select the workload)
TAG=${1:-cur}; LIB=${2:-diffus_amd/libdiffus_hip.so}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/run_step.py $LIB > $OUT/trace.log 2>&1
echo "trace rc=$?"
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/run_step.py $LIB > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    tmp = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]
        tmp[(n, r["Counter_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    for (n, c), v in tmp.items():
        g = max(x for x, _ in v); vals = sorted(t for x, t in v if x == g); res[n][c] = vals[len(vals) // 2]
dur = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]
        dur[n] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
tot = 0
for n in ("render_bwd_kernel", "scatter_patch_kernel", "gradbuf_flush_kernel"):
    c = res.get(n, {})
    w = c.get("SQ_WAVES", 0)
    print("== %-22s avg %.2f us (%d calls)" % (n, *dur.get(n, (0, 0))))
    for k, v in sorted(c.items()):
        print("   %-22s %14.0f   per wave %10.1f" % (k, v, v / w if w else 0))
    tot += c.get("SQ_INSTS_VALU", 0)
print("whole-step VALU wave-instructions: %.2f M" % (tot / 1e6))
PY
