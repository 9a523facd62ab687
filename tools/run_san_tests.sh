#!/bin/bash
# CPU tests of the host code (symbolic analysis, orderings, MatrixMarket parser, IBD builder, C-ABI argument checks) on the
# AddressSanitizer + UBSan build of the library (make -C scilmm_amd/csrc san).  Python itself is not instrumented, so the
# ASAN runtime is preloaded and leak checking (CPython "leaks" by design) is off.  No GPU is touched.
set -e
cd "$(dirname "$0")/.."
make -s -C scilmm_amd/csrc san
export SCILMM_HIP_LIB=$PWD/scilmm_amd/csrc/libscilmm_hip_san.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:verify_asan_link_order=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export SCILMM_NO_TORCH=1
exec python -m pytest tests/test_symbolic.py tests/test_harness.py tests/test_capi.py -q -m "not gpu" -p no:cacheprovider "$@"
