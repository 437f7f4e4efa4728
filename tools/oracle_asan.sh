#!/bin/bash
# The CPU oracle (oracle/vis_oracle.c) rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer and the golden-vector tests run
# against it (CPU box; the GPU pool refuses sanitizer runs).  -> profiles/rNN_oracle_asan.txt
set -euo pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
make -s -C "$root/oracle" asan
cd "$root"
ORACLE_LIB="$root/oracle/_build/liboracle_asan.so" LD_PRELOAD="$(gcc -print-file-name=libasan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4 python -m pytest tests/test_oracle_golden.py -q -x -p no:cacheprovider 2>&1 | tail -15
