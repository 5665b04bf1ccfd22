#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): every fuzzer once, logs under gpurun_out/fuzz/ (progress lines keep the run from looking hung).
#   gpurun --timeout 1100 -- 'bash tools/run_fuzzers.sh [first_seed]'
# About 12 minutes in all with the default counts.  What the round-5 campaigns found: profiles/r05_fuzz_*.txt, DESIGN facts 44-45.
set -u
F=${1:-100000}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/fuzz
mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 600 "$@" > $OUT/$name.txt 2>&1; echo "   rc=$? $(grep -v amdgpu $OUT/$name.txt | tail -1)"; }
run forward        python3 tools/fuzz_forward.py $F 6000
FUZZ_LONG=1   run forward_long   python3 tools/fuzz_forward.py $F 300
run one_pass       python3 tools/fuzz_one_pass.py 30000 $F
run echo           python3 tools/fuzz_echo.py $F 5000
run splat          python3 tools/fuzz_splat.py $F 1500
run aux            python3 tools/fuzz_aux.py $F 500
run slab           python3 tools/fuzz_slab.py $F 1200
FUZZ_PLANAR=1 run slab_planar    python3 tools/fuzz_slab.py $F 600
FUZZ_RANDOM=1 run slab_random    python3 tools/fuzz_slab.py $F 800
FUZZ_CROPPED=1 run slab_cropped  python3 tools/fuzz_slab.py $F 2000
FUZZ_LONG=1   run slab_long      python3 tools/fuzz_slab.py $F 100
run slab_medium    python3 tools/fuzz_slab_medium.py $F 40
