#!/bin/bash
# Host code (descriptor.cpp, plan_*.cpp, jit_planner.cpp, jit.cpp) under ThreadSanitizer on the CPU: tests/cpp/multi_device_test.cpp in its
# "host" mode -- 4 threads racing through the host-only entry points, the thread-local error messages, the runtime
# compiler (hiprtc for gfx950 needs no device), its process cache and the on-disk cache (cold directory).  Device
# objects from build/csrc; sanitizers stay on the CPU build (GPU sanitizers are not available on this pool).
# Run `make -C portfft_amd/csrc` first.   usage: tools/sanitize_tsan.sh [outdir]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/pfa_tsan}
rm -rf "$OUT"; mkdir -p "$OUT/cache"
SAN="-fsanitize=thread -fno-omit-frame-pointer -g"
HIPCC=/opt/rocm/bin/hipcc
cd "$ROOT/portfft_amd/csrc"
for f in descriptor plan_core plan_global plan_nd plan_exec jit_planner jit; do $HIPCC -O1 -std=c++17 -fPIC $SAN -I../../build/csrc -c $f.cpp -o "$OUT/$f.o" 2>/dev/null & done
wait
$HIPCC -shared -fPIC $SAN --offload-arch=gfx950 -o "$OUT/libportfft_amd.so" ../../build/csrc/kernels_*.o "$OUT"/descriptor.o "$OUT"/plan_core.o "$OUT"/plan_global.o "$OUT"/plan_nd.o "$OUT"/plan_exec.o "$OUT"/jit_planner.o "$OUT"/jit.o -ldl 2>/dev/null
$HIPCC -std=c++17 -O1 -pthread $SAN --offload-arch=gfx950 -I "$ROOT/include" "$ROOT/tests/cpp/multi_device_test.cpp" -L"$OUT" -lportfft_amd -Wl,-rpath,"$OUT" -o "$OUT/multi_device_test" 2>/dev/null
cd "$OUT"
# (the HIP runtime and hiprtc themselves are not instrumented: reports are restricted to frames of this library)
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" PFFT_JIT_CACHE_DIR="$OUT/cache" ./multi_device_test host 2>&1 | tee "$OUT/tsan.log" | grep -c "WARNING: ThreadSanitizer" | sed 's/^/ThreadSanitizer warnings: /' || true
grep -A12 "WARNING: ThreadSanitizer" "$OUT/tsan.log" | grep -E "pfa::|pfft_|portfft" | sort | uniq -c | head -20 || true
tail -2 "$OUT/tsan.log"
