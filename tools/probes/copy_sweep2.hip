// Scratch: does the per-lane burst depth / load-store interleave matter for streaming 32 KiB rows?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using b64_t = decltype(__builtin_amdgcn_raw_buffer_load_b64(std::declval<__amdgpu_buffer_rsrc_t>(), 0u, 0u, 0));

// MODE 0: per row: [load CHUNK, store CHUNK] x (16/CHUNK)
// MODE 1: persistent, pipelined: loads of row i+1 interleaved 1:1 with stores of row i
// MODE 2: persistent, pipelined: all loads of row i+1, then all stores of row i
template <int CHUNK, int MODE, int AUX>
__global__ __launch_bounds__(256) void rows_copy(const char* __restrict__ in, char* __restrict__ out, long long rows) {
  constexpr int ROWB = 32768, NACC = 16, VEC = 8, WG = 256;
  unsigned off = threadIdx.x * VEC;
  if constexpr (MODE == 0) {
    for (long long b = blockIdx.x; b < rows; b += gridDim.x) {
      auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in + b * ROWB), 0, ROWB, 0x00020000);
      auto rout = __builtin_amdgcn_make_buffer_rsrc(out + b * ROWB, 0, ROWB, 0x00020000);
#pragma unroll
      for (int c = 0; c < NACC; c += CHUNK) {
        b64_t v[CHUNK];
#pragma unroll
        for (int t = 0; t < CHUNK; ++t) v[t] = __builtin_amdgcn_raw_buffer_load_b64(rin, off, (c + t) * WG * VEC, AUX);
#pragma unroll
        for (int t = 0; t < CHUNK; ++t) __builtin_amdgcn_raw_buffer_store_b64(v[t], rout, off, (c + t) * WG * VEC, AUX);
      }
    }
  } else {
    long long b = blockIdx.x;
    if (b >= rows) return;
    b64_t cur[NACC], nxt[NACC];
    {
      auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in + b * ROWB), 0, ROWB, 0x00020000);
#pragma unroll
      for (int t = 0; t < NACC; ++t) cur[t] = __builtin_amdgcn_raw_buffer_load_b64(rin, off, t * WG * VEC, AUX);
    }
    for (; b < rows; b += gridDim.x) {
      long long bn = b + gridDim.x;
      // out-of-range next row: zero-size descriptor drops the loads
      auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in + (bn < rows ? bn : 0) * ROWB), 0, bn < rows ? ROWB : 0, 0x00020000);
      auto rout = __builtin_amdgcn_make_buffer_rsrc(out + b * ROWB, 0, ROWB, 0x00020000);
      if constexpr (MODE == 1) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) {
          nxt[t] = __builtin_amdgcn_raw_buffer_load_b64(rin, off, t * WG * VEC, AUX);
          __builtin_amdgcn_raw_buffer_store_b64(cur[t], rout, off, t * WG * VEC, AUX);
        }
      } else {
#pragma unroll
        for (int t = 0; t < NACC; ++t) nxt[t] = __builtin_amdgcn_raw_buffer_load_b64(rin, off, t * WG * VEC, AUX);
#pragma unroll
        for (int t = 0; t < NACC; ++t) __builtin_amdgcn_raw_buffer_store_b64(cur[t], rout, off, t * WG * VEC, AUX);
      }
#pragma unroll
      for (int t = 0; t < NACC; ++t) cur[t] = nxt[t];
    }
  }
}

static char *d_in, *d_out;
static long long g_rows;
template <int CHUNK, int MODE, int AUX>
void run(int grid_mode, int cus) {
  long long grid = grid_mode == 0 ? g_rows : (long long)grid_mode * cus;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto k = rows_copy<CHUNK, MODE, AUX>;
  float best = 1e9, tot = 0;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), 0, 0, d_in, d_out, g_rows);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) { tot += ms; if (ms < best) best = ms; }
  }
  double bytes = 2.0 * g_rows * 32768;
  printf("chunk=%-2d mode=%d aux=%d grid=%-6lld  avg %.4f ms %.2f TB/s   best %.4f ms %.2f TB/s\n", CHUNK, MODE, AUX, grid, tot / 5,
         bytes / (tot / 5) * 1e-9, best, bytes / best * 1e-9);
}
int main() {
  g_rows = 65536;
  size_t bytes = (size_t)g_rows * 32768;
  CK(hipMalloc(&d_in, bytes)); CK(hipMalloc(&d_out, bytes));
  CK(hipMemset(d_in, 1, bytes)); CK(hipMemset(d_out, 2, bytes));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  for (int gm : {0, 4, 8}) {
    run<1, 0, 2>(gm, cus); run<2, 0, 2>(gm, cus); run<4, 0, 2>(gm, cus); run<8, 0, 2>(gm, cus); run<16, 0, 2>(gm, cus);
    run<16, 0, 0>(gm, cus);
    if (gm) { run<16, 1, 2>(gm, cus); run<16, 2, 2>(gm, cus); run<16, 1, 0>(gm, cus); }
  }
  for (int gm : {2, 3, 5, 6}) { run<16, 0, 2>(gm, cus); run<16, 1, 2>(gm, cus); run<16, 2, 2>(gm, cus); }
  return 0;
}
