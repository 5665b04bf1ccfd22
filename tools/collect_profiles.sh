#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes (no --pmc together with
# any other trace domain) for ONE bench workload.  Output lands in gpurun_out/prof_$TAG (scratch); summaries are
# copied to profiles/ by tools/summarise_pmc.py afterwards (in the build container).
# Usage: tools/collect_profiles.sh TAG [bench args that define the workload...]
set -u
TAG=${1:-r03_c3}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
COMMON="--no-cpu-baseline --no-callers --no-scaling-legs --no-verify --eager $*"
python3 -c "import sys,json; sys.argv=['bench.py']+'$*'.split(); import bench; print(json.dumps(bench.workload_key(bench.parse_args())))" > $OUT/workload.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 30 --warmup 5 $COMMON > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
i=0
for C in "FETCH_SIZE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
         "WRITE_SIZE TCC_EA0_ATOMIC_sum SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 6 --warmup 2 $COMMON > $OUT/bench_pmc_$i.json 2> $OUT/pmc_$i.err
  echo "pmc pass $i rc=$?"
done
find $OUT -name "*.csv" | wc -l
