// Stand-alone kernel exploration harness (not part of the product, not used by tests or bench.py):
// times copy kernels (the achievable-HBM yardstick) and work-group FFT kernel variants on the C2 shape
// (fp32, N=4096, batch 65536) and checks a few batches against a double-precision DFT on the host.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probes/proto_c2.hip -o tools/probes/proto_c2 && tools/proto_c2
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#include "../portfft_amd/csrc/stockham_wg.hpp"

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

using namespace pfa;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void fill_kernel(float* p, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    z ^= z >> 31;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 29;
    p[i] = (float)((z >> 40) & 0xFFFFFF) * (2.0f / 16777216.0f) - 1.0f;
  }
}

template <typename V, bool NT>
__global__ void copy_kernel(const V* __restrict__ in, V* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    if constexpr (NT) {
      V v = __builtin_nontemporal_load(&in[i]);
      __builtin_nontemporal_store(v, &out[i]);
    } else {
      out[i] = in[i];
    }
  }
}

// copy with the FFT kernel's access shape: each work-group moves one 32 KiB row, 16 x 8B per lane at stride 256
template <int UNROLL>
__global__ __launch_bounds__(256) void copy_rows_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                        long long rows) {
  for (long long b = blockIdx.x; b < rows; b += gridDim.x) {
    float2 v[UNROLL];
#pragma unroll
    for (int t = 0; t < UNROLL; ++t) v[t] = in[b * (256 * UNROLL) + threadIdx.x + t * 256];
#pragma unroll
    for (int t = 0; t < UNROLL; ++t) out[b * (256 * UNROLL) + threadIdx.x + t * 256] = v[t];
  }
}

template <typename F>
float time_ms(F&& f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / reps;
}

template <typename Seq, typename T>
std::vector<cx<T>> make_twiddles() {
  std::vector<cx<T>> tw(Seq::tw_total > 0 ? Seq::tw_total : 1);
  for (int p = 1; p < Seq::count; ++p) {
    const int R = Seq::r[p], Ns = Seq::ns(p);
    for (int t = 1; t < R; ++t) {
      for (int q = 0; q < Ns; ++q) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)(t * q) / (long double)(Ns * R);
        tw[Seq::tw_off(p) + (t - 1) * Ns + q] = {(T)cosl(a), (T)sinl(a)};
      }
    }
  }
  return tw;
}

static const float2* g_in;
static float2* g_out;
static long long g_batch;
static std::vector<std::complex<double>> g_ref[3];
static long long g_ref_b[3];

template <typename Cfg>
void run_variant(const char* name, int wgs_per_cu, int num_cus) {
  using T = typename Cfg::T;
  auto tw = make_twiddles<typename Cfg::Seq, T>();
  cx<T>* d_tw;
  CK(hipMalloc(&d_tw, tw.size() * sizeof(cx<T>)));
  CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  auto kern = stockham_wg_kernel<Cfg, false>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, Cfg::WG, Cfg::LDS_BYTES));
  const long long nfft = g_batch * 4096 / Cfg::N;
  long long groups = (nfft + Cfg::FPW - 1) / Cfg::FPW;
  long long grid = wgs_per_cu > 0 ? std::min<long long>(groups, (long long)wgs_per_cu * num_cus) : groups;
  CK(hipMemset(g_out, 0, (size_t)g_batch * Cfg::N * sizeof(cx<T>)));
  float ms = time_ms(
      [&] {
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cfg::WG), Cfg::LDS_BYTES, 0, (const cx<T>*)g_in,
                           (cx<T>*)g_out, d_tw, nfft, (T)1);
      },
      10);
  // verify
  double worst = 0;
  std::vector<std::complex<float>> h(Cfg::N);
  for (int k = 0; k < 3 && Cfg::N == 4096; ++k) {
    CK(hipMemcpy(h.data(), g_out + g_ref_b[k] * Cfg::N, Cfg::N * sizeof(cx<T>), hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (int i = 0; i < Cfg::N; ++i) {
      std::complex<double> d = std::complex<double>(h[i]) - g_ref[k][i];
      num += std::norm(d);
      den += std::norm(g_ref[k][i]);
    }
    worst = std::max(worst, std::sqrt(num / den));
  }
  double bytes = 2.0 * nfft * Cfg::N * sizeof(cx<T>);
  double flops = 5.0 * Cfg::N * std::log2((double)Cfg::N) * nfft;
  printf("%-34s grid=%-6lld occ=%d lds=%-6zu  %.4f ms  %.2f TB/s  %.1f TFLOP/s  relL2=%.2e\n", name, grid, occ,
         Cfg::LDS_BYTES, ms, bytes / ms * 1e-9, flops / ms * 1e-9, worst);
  CK(hipFree(d_tw));
}

int main(int argc, char** argv) {
  const int N = 4096;
  g_batch = argc > 1 ? atoll(argv[1]) : 65536;
  size_t elems = (size_t)g_batch * N;
  float2 *d_in, *d_out;
  CK(hipMalloc(&d_in, elems * sizeof(float2)));
  CK(hipMalloc(&d_out, elems * sizeof(float2)));
  g_in = d_in;
  g_out = d_out;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs=%d LDS/block=%zu clock=%d MHz memclk=%d\n", prop.name, cus, prop.sharedMemPerBlock,
         prop.clockRate / 1000, prop.memoryClockRate / 1000);
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (float*)d_in, elems * 2);
  CK(hipDeviceSynchronize());

  // host reference for 3 batches: O(N^2) double DFT
  g_ref_b[0] = 0;
  g_ref_b[1] = g_batch / 2 + 1;
  g_ref_b[2] = g_batch - 1;
  {
    std::vector<std::complex<float>> h(N);
    std::vector<std::complex<double>> w(N);
    for (int i = 0; i < N; ++i) w[i] = std::polar(1.0, -2.0 * M_PI * i / N);
    for (int k = 0; k < 3; ++k) {
      CK(hipMemcpy(h.data(), d_in + g_ref_b[k] * N, N * sizeof(float2), hipMemcpyDeviceToHost));
      g_ref[k].resize(N);
      for (int o = 0; o < N; ++o) {
        std::complex<double> s = 0;
        for (int i = 0; i < N; ++i) s += std::complex<double>(h[i]) * w[(size_t)o * i % N];
        g_ref[k][o] = s;
      }
    }
  }

  double bytes = 2.0 * elems * sizeof(float2);
  // ---- copy yardsticks ----
  for (int blocks_per_cu : {4, 8, 16}) {
    int grid = cus * blocks_per_cu;
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL((copy_kernel<v4f, false>), dim3(grid), dim3(256), 0, 0, (const v4f*)d_in, (v4f*)d_out, elems / 2); }, 10);
    printf("copy float4     grid=%5d  %.4f ms  %.2f TB/s\n", grid, ms, bytes / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL((copy_kernel<v4f, true>), dim3(grid), dim3(256), 0, 0, (const v4f*)d_in, (v4f*)d_out, elems / 2); }, 10);
    printf("copy float4 nt  grid=%5d  %.4f ms  %.2f TB/s\n", grid, ms, bytes / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL((copy_kernel<v2f, false>), dim3(grid), dim3(256), 0, 0, (const v2f*)d_in, (v2f*)d_out, elems); }, 10);
    printf("copy float2     grid=%5d  %.4f ms  %.2f TB/s\n", grid, ms, bytes / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL((copy_kernel<v2f, true>), dim3(grid), dim3(256), 0, 0, (const v2f*)d_in, (v2f*)d_out, elems); }, 10);
    printf("copy float2 nt  grid=%5d  %.4f ms  %.2f TB/s\n", grid, ms, bytes / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL((copy_rows_kernel<16>), dim3(grid), dim3(256), 0, 0, (const float2*)d_in, (float2*)d_out, g_batch); }, 10);
    printf("copy rows16x8B  grid=%5d  %.4f ms  %.2f TB/s\n", grid, ms, bytes / ms * 1e-9);
  }
  {
    float ms = time_ms([&] { hipLaunchKernelGGL((copy_rows_kernel<16>), dim3((unsigned)g_batch), dim3(256), 0, 0, (const float2*)d_in, (float2*)d_out, g_batch); }, 10);
    printf("copy rows16x8B  grid=batch  %.4f ms  %.2f TB/s\n", ms, bytes / ms * 1e-9);
    ms = time_ms([&] { CK(hipMemcpyAsync(d_out, d_in, elems * sizeof(float2), hipMemcpyDeviceToDevice, 0)); }, 10);
    printf("hipMemcpy D2D   %.4f ms  %.2f TB/s\n", ms, bytes / ms * 1e-9);
  }

  // ---- FFT variants: interleaved rounds, medians ----
  using S16 = radix_list<16, 16, 16>;
  using CfgR = wg_cfg<float, S16, 256, 1, 16, 1, TW_REGS, 4, 2>;
  using CfgG = wg_cfg<float, S16, 256, 1, 16, 1, TW_GLOBAL, 4, 2>;
  auto tw = make_twiddles<S16, float>();
  cx<float>* d_tw;
  CK(hipMalloc(&d_tw, tw.size() * sizeof(cx<float>)));
  CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cx<float>), hipMemcpyHostToDevice));
  auto kr = stockham_wg_kernel<CfgR, false>;
#ifdef PF_TWR
  using CfgP = wg_cfg<float, S16, 256, 1, 16, 1, TW_REGS, 3, 2>;
#else
  using CfgP = wg_cfg<float, S16, 256, 1, 16, 1, TW_GLOBAL, 4, 2>;
#endif
  auto kg = stockham_wg_prefetch_kernel<CfgP, false>;
  CK(hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgR::LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)kg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgG::LDS_BYTES));
  const int grids[] = {512, 768, 1024, 1280, 2048, 4096, 8192, 12288, 16384, 24576, 32768};
  const int NG = sizeof(grids) / sizeof(grids[0]);
  std::vector<std::vector<float>> tr(NG), tg(NG);
  std::vector<float> tcopy;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timed = [&](auto&& launch) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
  };
  for (int round = 0; round < 9; ++round) {
    for (int gi = 0; gi < NG; ++gi) {
      float a = timed([&] { hipLaunchKernelGGL(kr, dim3(grids[gi]), dim3(256), CfgR::LDS_BYTES, 0, (const cx<float>*)g_in, (cx<float>*)g_out, d_tw, g_batch, 1.0f); });
      float b = timed([&] { hipLaunchKernelGGL(kg, dim3(grids[gi]), dim3(256), CfgG::LDS_BYTES, 0, (const cx<float>*)g_in, (cx<float>*)g_out, d_tw, g_batch, 1.0f); });
      if (round) { tr[gi].push_back(a); tg[gi].push_back(b); }
    }
    float c = timed([&] { hipLaunchKernelGGL((copy_rows_kernel<16>), dim3((unsigned)g_batch), dim3(256), 0, 0, (const float2*)d_in, (float2*)d_out, g_batch); });
    if (round) tcopy.push_back(c);
  }
  auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mn = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[0]; };
  printf("copy rows (no nt) median %.4f ms %.2f TB/s\n", med(tcopy), bytes / med(tcopy) * 1e-9);
  for (int gi = 0; gi < NG; ++gi) {
    printf("grid %6d  twR median %.4f ms %.2f TB/s (min %.4f)   PF median %.4f ms %.2f TB/s (min %.4f)\n", grids[gi], med(tr[gi]), bytes / med(tr[gi]) * 1e-9, mn(tr[gi]), med(tg[gi]), bytes / med(tg[gi]) * 1e-9, mn(tg[gi]));
  }
  return 0;
}
