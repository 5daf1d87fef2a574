#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer job for the CPU side (the GPU pool has no sanitizer runs): builds
# oracle/mrec_oracle.c with -fsanitize=address,undefined and runs the CPU test suite on it -- the oracle's own tests, the
# multi-rank host logic (which drives the oracle through the same ctypes front-end the GPU tests use) and the C-ABI argument
# validation of libmrec_hip.so (bad pointers / sizes must come back as error codes before any launch).
set -e
cd "$(dirname "$0")/.."
make -s -C oracle -B libmrec_oracle_san.so
export MREC_ORACLE_SANITIZE=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:handle_segv=0
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
python -m pytest tests/test_oracle.py tests/test_abi.py tests/test_wide_deep_dist.py tests/test_rec_model.py tests/test_criteo.py -q -m "not gpu" -x "$@"
