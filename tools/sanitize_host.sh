#!/bin/bash
# Host code (descriptor.cpp, plan_*.cpp, jit_planner.cpp, jit.cpp) under AddressSanitizer + UBSan on the CPU: the planner invariants test
# and the whole `-m "not gpu"` suite against a sanitized build of the library in /tmp (device objects from build/csrc;
# GPU sanitizers are not available on this pool).  Run `make -C portfft_amd/csrc` first.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/pfa_san}
mkdir -p "$OUT"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
HIPCC=/opt/rocm/bin/hipcc
cd "$ROOT/portfft_amd/csrc"
for f in descriptor plan_core plan_global plan_nd plan_exec jit_planner jit; do $HIPCC -O1 -std=c++17 -fPIC $SAN -I../../build/csrc -c $f.cpp -o "$OUT/$f.o" 2>/dev/null & done
wait
$HIPCC -shared -fPIC $SAN --offload-arch=gfx950 -o "$OUT/libportfft_amd.so" ../../build/csrc/kernels_*.o "$OUT"/descriptor.o "$OUT"/plan_core.o "$OUT"/plan_global.o "$OUT"/plan_nd.o "$OUT"/plan_exec.o "$OUT"/jit_planner.o "$OUT"/jit.o -ldl 2>/dev/null
$HIPCC -std=c++17 -O1 $SAN "$ROOT/tests/cpp/jit_planner_test.cpp" -L"$OUT" -lportfft_amd -Wl,-rpath,"$OUT" -o "$OUT/jit_planner_test" 2>/dev/null
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
( cd "$OUT" && ./jit_planner_test compile 2>&1 | grep -i "runtime error\|Sanitizer\|jit planner OK" )
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
# (test_missing_library_fails_loudly cannot hold while PORTFFT_AMD_LIBRARY points at a library)
PORTFFT_AMD_LIBRARY="$OUT/libportfft_amd.so" LD_PRELOAD="$RT" python -m pytest tests -q -m "not gpu" -s \
  --deselect tests/test_host_api.py::test_missing_library_fails_loudly 2>&1 | grep -i "runtime error\|Sanitizer\|passed\|failed"
