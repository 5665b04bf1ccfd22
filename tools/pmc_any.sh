#!/bin/bash
# GPU box: arbitrary PMC passes over tools/run_step.py (diagnostic).  Usage: tools/pmc_any.sh TAG LIB "C1 C2 ..." ["C3 ..."]...
TAG=$1; LIB=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcany_$TAG; mkdir -p $OUT
i=0
for C in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/run_step.py $LIB > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? ($C)"
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
for n in ("render_bwd_kernel", "scatter_patch_kernel"):
    print("==", n)
    for k, v in sorted(res.get(n, {}).items()):
        print("   %-30s %14.0f" % (k, v))
PY
