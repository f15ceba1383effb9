// Scratch: is the 4 KiB-per-work-group copy fast because each XCD keeps talking to the same HBM stack?
// Work-group i copies granule (i + shift) [mod n]; with round-robin work-group -> XCD placement the shift decides
// which granule residue class (mod 8) every XCD touches.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
template <int GRAN>  // bytes per work-group
__global__ __launch_bounds__(256) void copy_gran(const char* __restrict__ in, char* __restrict__ out, long long n, int shift, int scatter) {
  long long i = blockIdx.x;
  long long row = scatter ? ((i % 8) * (n / 8) + i / 8) : (i + shift) % n;   // scatter: XCD k gets a contiguous eighth
  const v4f* src = reinterpret_cast<const v4f*>(in + row * GRAN);
  v4f* dst = reinterpret_cast<v4f*>(out + row * GRAN);
#pragma unroll
  for (int k = 0; k < GRAN / 4096; ++k) {
    v4f v = __builtin_nontemporal_load(&src[threadIdx.x + k * 256]);
    __builtin_nontemporal_store(v, &dst[threadIdx.x + k * 256]);
  }
}
template <int GRAN> void run(const char* in, char* out, size_t bytes, int shift, int scatter) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  long long n = bytes / GRAN;
  std::vector<float> t;
  for (int r = 0; r < 7; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((copy_gran<GRAN>), dim3((unsigned)n), dim3(256), 0, 0, in, out, n, shift, scatter);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  printf("gran=%-6d shift=%d scatter=%d  median %.4f ms  %.2f TB/s\n", GRAN, shift, scatter, t[t.size()/2], 2.0 * bytes / t[t.size()/2] * 1e-9);
}
int main() {
  const size_t bytes = (size_t)2 << 30;
  char *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 2, bytes));
  printf("in %p out %p\n", in, out);
  for (int s = 0; s < 9; ++s) run<4096>(in, out, bytes, s, 0);
  run<4096>(in, out, bytes, 0, 1);
  for (int s : {0, 1, 3}) run<8192>(in, out, bytes, s, 0);
  for (int s : {0, 1}) run<32768>(in, out, bytes, s, 0);
  run<32768>(in, out, bytes, 0, 1);
  return 0;
}
