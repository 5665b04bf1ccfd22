#!/bin/bash
# The CPU oracle under AddressSanitizer + UBSan (CPU only: GPU sanitizers are not available on this pool).  Builds
# oracle/_build/liboracle_san.so and runs the oracle's CPU tests on it (the launcher / gloo tests spawn interpreters and are left out).
set -eu
cd "$(dirname "$0")/.."
make -s -C oracle sanitize
export DIFFUS_ORACLE_SO=$PWD/oracle/_build/liboracle_san.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
python -m pytest tests -x -q -m "not gpu" -k "oracle or golden or ill_conditioned or conditioning" "$@"
