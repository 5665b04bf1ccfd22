#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for
# the bench workload.  Output lands in gpurun_out/prof_$TAG (scratch); summaries are copied to
# profiles/ by tools/summarise_pmc.py afterwards (in the build container).
# Usage: tools/collect_profiles.sh TAG [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --eager $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_ATOMIC_sum"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --eager $* > $OUT/bench_$N.json 2> $OUT/pmc_$N.err
  echo "pmc $N rc=$?"
done
find $OUT -name "*.csv" | head -30
