#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU side: the oracle, the host encoder and synthesiser, the host
# side of the library (plans, pools, argument checks) and the C++ host headers.  Needs no GPU (device code is not sanitized).
#   tools/sanitize.sh [--keep]   builds the three sanitized artefacts, runs the CPU tests that exercise them, removes them
# Leak checking is off: the HIP runtime and the Python interpreter keep allocations until exit.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
make -C oracle asan
make -C indexr_amd/csrc asan
make -C tests/cpp host_test_asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
mkdir -p /tmp/ixr_asan
echo "== host_test --cpu under ASan + UBSan"
tests/cpp/host_test_asan --cpu /tmp/ixr_asan
echo "== python CPU tests with the sanitized oracle and library"
IXR_LIB=$ROOT/indexr_amd/libixrgpu_asan.so IXO_LIB=$ROOT/oracle/liboracle_asan.so LD_PRELOAD=$RT \
  python -m pytest tests/test_oracle.py tests/test_host_encoder.py tests/test_golden.py tests/test_select.py tests/test_scan_filter.py tests/test_handback.py tests/test_filter.py -q -m "not gpu" -x -p no:cacheprovider
echo "sanitizer run clean"
# the sanitized artefacts are for this run only (every gpurun call ships the tree's .so files)
[ "$1" = "--keep" ] || rm -rf indexr_amd/libixrgpu_asan.so indexr_amd/csrc/build_asan oracle/liboracle_asan.so tests/cpp/host_test_asan
