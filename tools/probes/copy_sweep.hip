// Scratch: which HBM access shapes reach the best streaming rate for 32 KiB rows (the C2 FFT's unit of work)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using b64_t = decltype(__builtin_amdgcn_raw_buffer_load_b64(std::declval<__amdgpu_buffer_rsrc_t>(), 0u, 0u, 0));
using b128_t = decltype(__builtin_amdgcn_raw_buffer_load_b128(std::declval<__amdgpu_buffer_rsrc_t>(), 0u, 0u, 0));

// ROWB bytes per row; WG threads; VEC bytes per access (8 or 16)
template <int ROWB, int WG, int VEC, int AUX_LD, int AUX_ST, bool CHUNKED>
__global__ __launch_bounds__(WG) void rows_copy(const char* __restrict__ in, char* __restrict__ out, long long rows) {
  constexpr int NACC = ROWB / (WG * VEC);
  long long per = (rows + gridDim.x - 1) / gridDim.x;
  long long b0 = CHUNKED ? blockIdx.x * per : blockIdx.x;
  long long b1 = CHUNKED ? (b0 + per < rows ? b0 + per : rows) : rows;
  long long bs = CHUNKED ? 1 : gridDim.x;
  for (long long b = b0; b < b1; b += bs) {
    auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in + b * ROWB), 0, ROWB, 0x00020000);
    auto rout = __builtin_amdgcn_make_buffer_rsrc(out + b * ROWB, 0, ROWB, 0x00020000);
    unsigned off = threadIdx.x * VEC;
    if constexpr (VEC == 8) {
      b64_t v[NACC];
#pragma unroll
      for (int t = 0; t < NACC; ++t) v[t] = __builtin_amdgcn_raw_buffer_load_b64(rin, off, t * WG * VEC, AUX_LD);
#pragma unroll
      for (int t = 0; t < NACC; ++t) __builtin_amdgcn_raw_buffer_store_b64(v[t], rout, off, t * WG * VEC, AUX_ST);
    } else {
      b128_t v[NACC];
#pragma unroll
      for (int t = 0; t < NACC; ++t) v[t] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, t * WG * VEC, AUX_LD);
#pragma unroll
      for (int t = 0; t < NACC; ++t) __builtin_amdgcn_raw_buffer_store_b128(v[t], rout, off, t * WG * VEC, AUX_ST);
    }
  }
}

static char *d_in, *d_out;
static long long g_rows;
template <int ROWB, int WG, int VEC, int AUX_LD, int AUX_ST, bool CHUNKED>
void run(int grid_mode, int cus) {
  const long long rows = g_rows * 32768 / ROWB;
  long long grid = grid_mode == 0 ? rows : (long long)grid_mode * cus;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto k = rows_copy<ROWB, WG, VEC, AUX_LD, AUX_ST, CHUNKED>;
  float best = 1e9, tot = 0;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(WG), 0, 0, d_in, d_out, rows);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) { tot += ms; if (ms < best) best = ms; }
  }
  double bytes = 2.0 * rows * ROWB;
  printf("row=%-6d wg=%-4d vec=%-2d ld=%d st=%d %s grid=%-6lld  avg %.4f ms %.2f TB/s   best %.4f ms %.2f TB/s\n", ROWB, WG, VEC, AUX_LD, AUX_ST,
         CHUNKED ? "chunk" : "inter", grid, tot / 5, bytes / (tot / 5) * 1e-9, best, bytes / best * 1e-9);
}
template <int ROWB, int WG, int VEC, bool CHUNKED>
void sweep_aux(int gm, int cus) {
  run<ROWB, WG, VEC, 0, 0, CHUNKED>(gm, cus);
  run<ROWB, WG, VEC, 2, 0, CHUNKED>(gm, cus);
  run<ROWB, WG, VEC, 0, 2, CHUNKED>(gm, cus);
  run<ROWB, WG, VEC, 2, 2, CHUNKED>(gm, cus);
}
int main() {
  g_rows = 65536;
  size_t bytes = (size_t)g_rows * 32768;
  CK(hipMalloc(&d_in, bytes)); CK(hipMalloc(&d_out, bytes));
  CK(hipMemset(d_in, 1, bytes)); CK(hipMemset(d_out, 2, bytes));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  for (int gm : {0, 4, 8}) {
    run<32768, 256, 8, 2, 2, false>(gm, cus);
    run<32768, 256, 16, 2, 2, false>(gm, cus);
    run<32768, 128, 16, 2, 2, false>(gm, cus);
    run<32768, 512, 8, 2, 2, false>(gm, cus);
    run<32768, 512, 16, 2, 2, false>(gm, cus);
    run<32768, 1024, 16, 2, 2, false>(gm, cus);
    run<32768, 1024, 8, 2, 2, false>(gm, cus);
    run<4096, 256, 16, 2, 2, false>(gm, cus);
    run<8192, 256, 16, 2, 2, false>(gm, cus);
    run<8192, 256, 8, 2, 2, false>(gm, cus);
    run<16384, 256, 16, 2, 2, false>(gm, cus);
    run<16384, 256, 8, 2, 2, false>(gm, cus);
    run<65536, 256, 16, 2, 2, false>(gm, cus);
    run<65536, 512, 16, 2, 2, false>(gm, cus);
    run<65536, 512, 8, 2, 2, false>(gm, cus);
    run<131072, 1024, 16, 2, 2, false>(gm, cus);
  }
  return 0;
}
