// Yardstick for BASELINE config 5 (fp32 1024 x 1024 x 256, two-pass 2-D plan) in SPLIT_COMPLEX storage against
// the interleaved one: plain copies with the two passes' access shapes, nothing else.
//   pass 1 (rows2d): work-group b of a matrix loads the RC = 8 rows {M a + b} (1024 contiguous elements each) and stores the
//                    8 adjacent rows RC b + u -- one contiguous block of 8192 elements
//   pass 2 (strided, in place): groups of 64 adjacent columns x M = 128 rows at a pitch of RC n1 = 8192 elements
// Interleaved: 8-byte elements (pass 2: 512-byte segments).  Split: two planes of 4-byte scalars (pass 2: 256-byte segments
// per plane, twice the memory instructions).  nt loads and stores over 2 GiB (in + out), median of 5.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/copy_c5_split.hip -o build/copy_c5_split
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int N1 = 1024, N0 = 1024, RC = 8, M = N0 / RC;

template <bool SPLIT>
__global__ __launch_bounds__(512) void pass1(const float* in_re, const float* in_im, float* out_re, float* out_im, long long nmat) {
  const v2f* in = reinterpret_cast<const v2f*>(in_re);
  v2f* out = reinterpret_cast<v2f*>(out_re);
  for (long long g = blockIdx.x; g < nmat * M; g += gridDim.x) {
    const long long mat = g / M, b = g % M, base = mat * (long long)N0 * N1;
    v2f v[RC][N1 / 512];
#pragma unroll
    for (int a = 0; a < RC; ++a)
#pragma unroll
      for (int k = 0; k < N1 / 512; ++k) {
        const long long o = base + (long long)(M * a + b) * N1 + threadIdx.x + k * 512;
        if (SPLIT) { v[a][k].x = __builtin_nontemporal_load(in_re + o); v[a][k].y = __builtin_nontemporal_load(in_im + o); }
        else v[a][k] = __builtin_nontemporal_load(in + o);
      }
#pragma unroll
    for (int u = 0; u < RC; ++u)
#pragma unroll
      for (int k = 0; k < N1 / 512; ++k) {
        const long long o = base + (long long)(RC * b + u) * N1 + threadIdx.x + k * 512;
        if (SPLIT) { __builtin_nontemporal_store(v[u][k].x, out_re + o); __builtin_nontemporal_store(v[u][k].y, out_im + o); }
        else __builtin_nontemporal_store(v[u][k], out + o);
      }
  }
}

// (the plan's pass 2 works in place; a copy that stores what it loaded to the same address is dead code to the compiler,
//  so the yardstick reads `src` and writes the same positions of `re` / `im`: same shapes, same bytes)
template <bool SPLIT>
__global__ __launch_bounds__(512) void pass2(const float* sre, const float* sim, float* re, float* im, long long nmat) {
  v2f* io = reinterpret_cast<v2f*>(re);
  const v2f* src = reinterpret_cast<const v2f*>(sre);
  constexpr int COLS = 64, RPI = 512 / COLS, IT = M / RPI, PITCH = RC * N1, GPM = PITCH / COLS;
  const int c = threadIdx.x % COLS, r0 = threadIdx.x / COLS;
  for (long long g = blockIdx.x; g < nmat * GPM; g += gridDim.x) {
    const long long base = (g / GPM) * (long long)N0 * N1 + (g % GPM) * COLS + c;
    v2f v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const long long o = base + (long long)(r0 + i * RPI) * PITCH;
      if (SPLIT) { v[i].x = __builtin_nontemporal_load(sre + o); v[i].y = __builtin_nontemporal_load(sim + o); }
      else v[i] = __builtin_nontemporal_load(src + o);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const long long o = base + (long long)(r0 + i * RPI) * PITCH;
      if (SPLIT) { __builtin_nontemporal_store(v[i].x, re + o); __builtin_nontemporal_store(v[i].y, im + o); }
      else __builtin_nontemporal_store(v[i], io + o);
    }
  }
}

int main() {
  const long long nmat = 256;
  const size_t elems = (size_t)nmat * N0 * N1;
  float *a, *b;
  CK(hipMalloc(&a, elems * 8)); CK(hipMalloc(&b, elems * 8));
  CK(hipMemset(a, 1, elems * 8)); CK(hipMemset(b, 2, elems * 8));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); const int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto&& launch) {
    std::vector<float> t;
    for (int r = 0; r < 6; ++r) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("%-44s %.3f ms  %.2f TB/s\n", name, t[2], 2.0 * elems * 8 / (t[2] * 1e-3) / 1e12);
    CK(hipGetLastError());
    return t[2];
  };
  const unsigned g1 = 8 * cus, g2 = 8 * cus;
  const float i1 = timeit("pass 1, interleaved", [&] { hipLaunchKernelGGL(pass1<false>, dim3(g1), dim3(512), 0, 0, a, nullptr, b, nullptr, nmat); });
  const float s1 = timeit("pass 1, split planes", [&] { hipLaunchKernelGGL(pass1<true>, dim3(g1), dim3(512), 0, 0, a, a + elems, b, b + elems, nmat); });
  const float i2 = timeit("pass 2, interleaved (512-byte segments)", [&] { hipLaunchKernelGGL(pass2<false>, dim3(g2), dim3(512), 0, 0, b, nullptr, a, nullptr, nmat); });
  const float s2 = timeit("pass 2, split planes (256-byte segments)", [&] { hipLaunchKernelGGL(pass2<true>, dim3(g2), dim3(512), 0, 0, b, b + elems, a, a + elems, nmat); });
  const double bytes = 2.0 * elems * 8;
  printf("copy pair as a fraction of 8 TB/s on 1x bytes: interleaved %.3f, split %.3f (split / interleaved %.3f)\n",
         bytes / ((i1 + i2) * 1e-3) / 8e12, bytes / ((s1 + s2) * 1e-3) / 8e12, (i1 + i2) / (s1 + s2));
  return 0;
}
