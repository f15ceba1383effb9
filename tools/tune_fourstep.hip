// Pipeline tuner for the two stage kernels of the four-step (GLOBAL) tier, under the conditions of a plan's execute:
// N = n1 x n2, `batch` transforms run in Infinity-Cache-sized chunks (256 MiB of intermediate), stage A with the
// writer cache policy (store modifier W_N^(k1*c) from LDS tables), stage B with the reader policy, uniform(-1, 1)
// random data, every kernel timed with events inside the A, B, A, B ... sequence, grids by the library's rule.
// Stage-A and stage-B variants are measured independently (each against a fixed partner with the same intermediate
// layout) and every pair's result is compared with the production pair's.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTUNE_CASE=<case> tools/tune_fourstep.hip -o build/tune_fourstep_<case>
//   cases: 20 = fp32 2^20 (1024 x 1024), 18 = fp32 2^18 (512 x 512), 22 = fp32 2^22 (2048 x 2048),
//          16 = fp32 65536 (256 x 256), 120 = fp64 2^20 (C3)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include "probes/stockham_strided_hx.hpp"
#include "probes/stockham_strided_sfr.hpp"
#include "probes/stockham_strided_dg.hpp"
#include "../portfft_amd/csrc/kernels.hpp"
using namespace pfa;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

#ifndef TUNE_CASE
#define TUNE_CASE 20
#endif
#if TUNE_CASE == 120
using T = double;
constexpr long long N1 = 1024, N2 = 1024, BATCH = 64;
#elif TUNE_CASE == 116
using T = double;
constexpr long long N1 = 256, N2 = 256, BATCH = 1024;
#elif TUNE_CASE == 118
using T = double;
constexpr long long N1 = 512, N2 = 512, BATCH = 256;
#elif TUNE_CASE == 18
using T = float;
constexpr long long N1 = 512, N2 = 512, BATCH = 512;
#elif TUNE_CASE == 15
using T = float;
constexpr long long N1 = 128, N2 = 256, BATCH = 4096;
#elif TUNE_CASE == 22
using T = float;
constexpr long long N1 = 2048, N2 = 2048, BATCH = 32;
#elif TUNE_CASE == 16
using T = float;
constexpr long long N1 = 256, N2 = 256, BATCH = 2048;
#else
using T = float;
constexpr long long N1 = 1024, N2 = 1024, BATCH = 128;
#endif
constexpr long long N = N1 * N2;
constexpr int W = PFA_AUX_WRITER, RD = PFA_AUX_READER;

template <typename Seq>
cx<T>* make_twiddles() {
  std::vector<cx<T>> tw(Seq::tw_total > 0 ? Seq::tw_total : 1);
  for (int p = 1; p < Seq::count; ++p) {
    const int R = Seq::r[p], Ns = Seq::ns(p);
    for (int t = 1; t < R; ++t) for (int q = 0; q < Ns; ++q) {
      const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)(t * q) / (long double)(Ns * R);
      tw[Seq::tw_off(p) + (t - 1) * Ns + q] = {(T)cosl(a), (T)sinl(a)};
    }
  }
  cx<T>* d; CK(hipMalloc(&d, tw.size() * sizeof(cx<T>)));
  CK(hipMemcpy(d, tw.data(), tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  return d;
}

__global__ void fill_uniform(T* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + seed * 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
    p[i] = (T)((double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0);
  }
}

enum kind_t { K_PLAIN = 0, K_PREFETCH = 1, K_HX = 2, K_ROW_IN = 3, K_TIN = 4, K_NOSTW = 5 /* timing only: stage A without its store modifier */, K_PF_TIN = 6, K_SFR = 7, K_PF_TIN_LTW = 8 /* stage B carrying the modifier on its loads */, K_PF_NOSTW = 9 /* timing only */, K_DG = 10 /* stage A loading two groups' columns at once */ };
struct variant {
  std::string name;
  int kind, fpw, wg, gpw;
  bool tiled;       // group-major intermediate on the scratch side
  size_t lds;
  const void* fn;
  const void* tw;
  int r0, rlast;    // first / last radix (the tiled layout needs divisibility)
  std::function<void(unsigned, const strided_args&)> launch;
};
static std::vector<variant> g_a, g_b;
static int g_stw_levels, g_stw_shift;
static void* g_stw_tab;

template <typename Cfg, int KIND, bool STAGE_A>
void add(const char* name, bool tiled, int gpw, size_t extra_lds = 0) {
  constexpr int STW = (STAGE_A && KIND != K_NOSTW && KIND != K_PF_NOSTW) ? 1 : 0;
  const void* fn;
  size_t lds;
  if constexpr (KIND == K_PREFETCH || KIND == K_PF_NOSTW) { fn = (const void*)&stockham_strided_prefetch_kernel<Cfg, false, STW>; lds = strided_lds_bytes<Cfg>(); }
  else if constexpr (KIND == K_PF_TIN) { fn = (const void*)&stockham_strided_prefetch_kernel<Cfg, false, STW, 0, true>; lds = strided_lds_bytes<Cfg>(); }
  else if constexpr (KIND == K_SFR) { fn = (const void*)&stockham_strided_sfr_kernel<Cfg, false, STW>; lds = strided_sfr_lds_bytes<Cfg>(); }
  else if constexpr (KIND == K_PF_TIN_LTW) { fn = (const void*)&stockham_strided_prefetch_kernel<Cfg, false, 0, 0, true, 1>; lds = strided_lds_bytes<Cfg>() + ((size_t)g_stw_levels << g_stw_shift) * sizeof(cx<T>); }
  else if constexpr (KIND == K_HX) { fn = (const void*)&stockham_strided_hx_kernel<Cfg, false, STW>; lds = strided_hx_lds_bytes<Cfg>(); }
  else if constexpr (KIND == K_ROW_IN) { fn = (const void*)&stockham_strided_row_kernel<Cfg, false, true, false>; lds = strided_row_lds_bytes<Cfg>(); }
  else if constexpr (KIND == K_TIN) { fn = (const void*)&stockham_strided_kernel<Cfg, false, STW, 0, true>; lds = strided_lds_bytes<Cfg>(); }
  else if constexpr (KIND == K_DG) { fn = (const void*)&stockham_strided_dg_kernel<Cfg, false, STW>; lds = strided_lds_bytes<Cfg>(); }
  else { fn = (const void*)&stockham_strided_kernel<Cfg, false, STW>; lds = strided_lds_bytes<Cfg>(); }
  if (STAGE_A) lds += ((size_t)g_stw_levels << g_stw_shift) * sizeof(cx<T>);
  lds += extra_lds;  // (occupancy experiments: unused LDS that costs resident work-groups)
  CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  variant v;
  v.name = name; v.kind = KIND; v.fpw = Cfg::FPW; v.wg = Cfg::WG; v.gpw = gpw; v.tiled = tiled; v.lds = lds; v.fn = fn;
  v.tw = make_twiddles<typename Cfg::Seq>();
  v.r0 = Cfg::Seq::r[0]; v.rlast = Cfg::Seq::r[Cfg::NP - 1];
  v.launch = [lds](unsigned grid, const strided_args& a) {
    if constexpr (KIND == K_PREFETCH || KIND == K_PF_NOSTW) hipLaunchKernelGGL((stockham_strided_prefetch_kernel<Cfg, false, STW>), dim3(grid), dim3(Cfg::WG), lds, 0, a);
    else if constexpr (KIND == K_PF_TIN) hipLaunchKernelGGL((stockham_strided_prefetch_kernel<Cfg, false, STW, 0, true>), dim3(grid), dim3(Cfg::WG), lds, 0, a);
    else if constexpr (KIND == K_SFR) hipLaunchKernelGGL((stockham_strided_sfr_kernel<Cfg, false, STW>), dim3(grid), dim3(Cfg::WG), lds, 0, a);
    else if constexpr (KIND == K_PF_TIN_LTW) hipLaunchKernelGGL((stockham_strided_prefetch_kernel<Cfg, false, 0, 0, true, 1>), dim3(grid), dim3(Cfg::WG), lds, 0, a);
    else if constexpr (KIND == K_HX) hipLaunchKernelGGL((stockham_strided_hx_kernel<Cfg, false, STW>), dim3(grid), dim3(Cfg::WG), lds, 0, a);
    else if constexpr (KIND == K_ROW_IN) hipLaunchKernelGGL((stockham_strided_row_kernel<Cfg, false, true, false>), dim3(grid), dim3(Cfg::WG), lds, 0, a);
    else if constexpr (KIND == K_TIN) hipLaunchKernelGGL((stockham_strided_kernel<Cfg, false, STW, 0, true>), dim3(grid), dim3(Cfg::WG), lds, 0, a);
    else if constexpr (KIND == K_DG) hipLaunchKernelGGL((stockham_strided_dg_kernel<Cfg, false, STW>), dim3(grid), dim3(Cfg::WG), lds, 0, a);
    else hipLaunchKernelGGL((stockham_strided_kernel<Cfg, false, STW>), dim3(grid), dim3(Cfg::WG), lds, 0, a);
  };
  (STAGE_A ? g_a : g_b).push_back(v);
}
template <typename Cfg, int KIND> void addA(const char* name, bool tiled, int gpw, size_t x = 0) { add<Cfg, KIND, true>(name, tiled, gpw, x); }
template <typename Cfg, int KIND> void addB(const char* name, bool tiled, int gpw, size_t x = 0) { add<Cfg, KIND, false>(name, tiled, gpw, x); }

static int g_cus = 256;
static unsigned grid_of(const variant& v, long long groups) {  // plan_core.cpp persistent_grid
  int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v.fn, v.wg, v.lds));
  per_cu = std::max(per_cu, 1);
  const long long resident = (long long)per_cu * g_cus;
  long long grid = v.gpw <= 0 ? 2 * resident : (groups + v.gpw - 1) / v.gpw;
  grid = std::min(groups, std::max(grid, std::min<long long>(groups, 2 * resident)));
  return (unsigned)std::max<long long>(1, grid);
}

// strided_args of the two stages for `nb` transforms starting at user transform b0 (plan_global.cpp plan_global)
static strided_args args_a(const variant& v, const T* in, T* scratch, long long nb, int t_layout) {
  strided_args a{};
  a.in = in; a.out = scratch; a.tw = v.tw; a.total = nb * N2; a.inner = N2;
  a.in_dist_outer = N; a.out_dist_outer = N; a.in_stride = (unsigned)N2; a.out_stride = (unsigned)N2; a.in_fdist = 1; a.out_fdist = 1;
  a.scale = 1.0; a.stw_tab = g_stw_tab; a.stw_levels = g_stw_levels; a.stw_lshift = g_stw_shift; a.stw_cdiv = 1;
  if (t_layout > 0) { a.out_gdist = N1 * t_layout; a.out_stride = (unsigned)t_layout; a.out_fdist = 1; }
  return a;
}
static strided_args args_b(const variant& v, const T* scratch, T* out, long long nb, int t_layout) {
  strided_args a{};
  a.in = scratch; a.out = out; a.tw = v.tw; a.total = nb * N1; a.inner = N1;
  a.in_dist_outer = N; a.out_dist_outer = N; a.in_stride = 1; a.in_fdist = (unsigned)N2; a.out_stride = (unsigned)N1; a.out_fdist = 1;
  a.scale = 1.0; a.stw_tab = g_stw_tab; a.stw_levels = g_stw_levels; a.stw_lshift = g_stw_shift; a.stw_cdiv = 1;
  if (t_layout > 0) {
    int sh = 0; while ((1 << sh) < t_layout) ++sh;
    a.in_tile_shift = sh; a.in_stride = (unsigned)(N1 * t_layout); a.in_fdist = (unsigned)t_layout;
  }
  return a;
}

struct result { double a_us, b_us, total_ms; };
static result run_pair(const variant& va, const variant& vb, const T* in, T* scratch, T* out, int reps) {
  const size_t per = (size_t)N * sizeof(cx<T>);
  const long long chunk = std::max<long long>(1, std::min<long long>(BATCH, (long long)(((size_t)256 << 20) / per)));
  const int t_layout = va.tiled ? va.fpw : 0;
  const int nch = (int)((BATCH + chunk - 1) / chunk);
  std::vector<hipEvent_t> ev(2 * nch + 1);
  for (auto& e : ev) CK(hipEventCreate(&e));
  std::vector<double> ta, tb, tt;
  for (int rep = 0; rep <= reps; ++rep) {
    CK(hipEventRecord(ev[0]));
    for (int c = 0; c < nch; ++c) {
      const long long b0 = c * chunk, nb = std::min(chunk, BATCH - b0);
      const strided_args aa = args_a(va, in + 2 * b0 * N, scratch, nb, t_layout);
      va.launch(grid_of(va, (nb * N2) / (va.kind == K_DG ? 2 * va.fpw : va.fpw)), aa);
      CK(hipEventRecord(ev[2 * c + 1]));
      const strided_args ab = args_b(vb, scratch, out + 2 * b0 * N, nb, t_layout);
      vb.launch(grid_of(vb, (nb * N1) / vb.fpw), ab);
      CK(hipEventRecord(ev[2 * c + 2]));
    }
    CK(hipEventSynchronize(ev[2 * nch]));
    double a = 0, b = 0;
    for (int c = 0; c < nch; ++c) {
      float ms; CK(hipEventElapsedTime(&ms, ev[2 * c], ev[2 * c + 1])); a += ms;
      CK(hipEventElapsedTime(&ms, ev[2 * c + 1], ev[2 * c + 2])); b += ms;
    }
    if (rep) { ta.push_back(a / nch * 1e3); tb.push_back(b / nch * 1e3); tt.push_back(a + b); }
  }
  CK(hipGetLastError());
  for (auto& e : ev) CK(hipEventDestroy(e));
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  return {med(ta), med(tb), med(tt)};
}

static double compare(const T* d_out, const std::vector<T>& ref, size_t count) {
  std::vector<T> h(count);
  CK(hipMemcpy(h.data(), d_out, count * sizeof(T), hipMemcpyDeviceToHost));
  double num = 0, den = 0;
  for (size_t i = 0; i < count; ++i) { const double d = (double)h[i] - (double)ref[i]; num += d * d; den += (double)ref[i] * (double)ref[i]; }
  return std::sqrt(num / std::max(den, 1e-300));
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); g_cus = prop.multiProcessorCount;
  const size_t total = (size_t)BATCH * N * 2;  // scalars
  T *in, *out, *scratch;
  CK(hipMalloc(&in, total * sizeof(T))); CK(hipMalloc(&out, total * sizeof(T))); CK(hipMalloc(&scratch, (size_t)256 << 20));
  fill_uniform<<<4096, 256>>>(in, total, 7);
  CK(hipDeviceSynchronize());
  {  // store-modifier tables (plan_core.cpp store_table_shape: fewest levels within 16 KiB)
    int bits = 0; while ((1ll << bits) < N) ++bits;
    for (int l = 1; l <= 4; ++l) {
      const int sh = (bits + l - 1) / l;
      if (((size_t)l << sh) * sizeof(cx<T>) <= 16 * 1024) { g_stw_levels = l; g_stw_shift = sh; break; }
    }
    if (const char* e = getenv("TUNE_STW_LEVELS")) { g_stw_levels = atoi(e); g_stw_shift = (bits + g_stw_levels - 1) / g_stw_levels; }
    const long long per = 1ll << g_stw_shift;
    std::vector<cx<T>> tab((size_t)g_stw_levels * per);
    for (int l = 0; l < g_stw_levels; ++l) for (long long i = 0; i < per; ++i) {
      const long long m = (long long)(((unsigned long long)i << (l * g_stw_shift)) % (unsigned long long)N);
      const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)N;
      tab[l * per + i] = {(T)cosl(a), (T)sinl(a)};
    }
    CK(hipMalloc(&g_stw_tab, tab.size() * sizeof(tab[0]))); CK(hipMemcpy(g_stw_tab, tab.data(), tab.size() * sizeof(tab[0]), hipMemcpyHostToDevice));
    printf("N = %lld x %lld, batch %lld, %s, store-modifier tables: %d levels of %lld entries\n", N1, N2, BATCH, sizeof(T) == 4 ? "fp32" : "fp64", g_stw_levels, per);
  }
  using f = float; using d = double;
  (void)sizeof(f); (void)sizeof(d);
#if TUNE_CASE == 20
  // production: A = 16.8.8 on 1024 lanes (plain layout), B = the same entry's row-staged form
  addA<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, W>, K_PLAIN>("A 16.8.8 wg1024 fpw16 plain-layout (production)", false, 4);
  addB<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, RD>, K_ROW_IN>("B 16.8.8 wg1024 fpw16 row-staged (production)", false, 4);
  addA<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, W>, K_PLAIN>("A 16.8.8 wg1024 fpw16 tiled", true, 4);
  addB<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, RD>, K_TIN>("B 16.8.8 wg1024 fpw16 tiled TIN", true, 4);
  addB<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, RD>, K_PLAIN>("B 16.8.8 wg1024 fpw16 tiled", true, 4);
  addA<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, W>, K_PREFETCH>("A PF 16.8.8 wg1024 fpw16 tiled", true, 4);
  addB<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, RD>, K_PREFETCH>("B PF 16.8.8 wg1024 fpw16 tiled", true, 4);
  addB<strided_cfg<f, radix_list<32, 32>, 512, 16, 2, RD>, K_PREFETCH>("B PF 32.32 wg512 fpw16 tiled", true, 4);
  addB<strided_cfg<f, radix_list<32, 32>, 512, 16, 2, RD>, K_TIN>("B 32.32 wg512 fpw16 tiled TIN", true, 4);
  addA<strided_cfg<f, radix_list<16, 8, 8>, 512, 16, 4, W>, K_HX>("A HX 16.8.8 wg512(32pt) fpw16 tiled 2/CU", true, 4);
  addB<strided_cfg<f, radix_list<16, 8, 8>, 512, 16, 4, RD>, K_HX>("B HX 16.8.8 wg512(32pt) fpw16 tiled 2/CU", true, 4);
  addB<strided_cfg<f, radix_list<32, 32>, 512, 16, 4, RD>, K_HX>("B HX 32.32 wg512(32pt) fpw16 tiled 2/CU", true, 4);
  addA<sfr_cfg<f, radix_list<2, 8, 8, 8>, 512, 16, 4, W>, K_SFR>("A SFR 2.8.8.8 wg512 fpw16 tiled 2/CU", true, 4);
  addA<sfr_cfg<f, radix_list<2, 8, 8, 8>, 512, 16, 4, W>, K_SFR>("A SFR 2.8.8.8 wg512 fpw16 tiled 2/CU gpw2", true, 2);
  addB<sfr_cfg<f, radix_list<2, 8, 8, 8>, 512, 16, 4, RD>, K_SFR>("B SFR 2.8.8.8 wg512 fpw16 tiled 2/CU", true, 4);
  addA<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, W>, K_NOSTW>("A 16.8.8 wg1024 fpw16 tiled WITHOUT stw (timing only)", true, 4);
  addA<strided_cfg<f, radix_list<32, 32>, 512, 16, 2, W>, K_NOSTW>("A 32.32 wg512 fpw16 tiled WITHOUT stw (timing only)", true, 4);
  addA<strided_cfg<f, radix_list<32, 32>, 512, 16, 2, W>, K_PLAIN>("A 32.32 wg512 fpw16 tiled", true, 4);
  addA<strided_cfg<f, radix_list<16, 8, 8>, 512, 16, 2, W>, K_PLAIN>("A 16.8.8 wg512(32pt) fpw16 tiled", true, 8);
  addA<strided_cfg<f, radix_list<16, 8, 8>, 512, 16, 2, W>, K_PREFETCH>("A PF 16.8.8 wg512(32pt) fpw16 tiled", true, 8);
  addA<strided_cfg<f, radix_list<32, 32>, 512, 16, 1, W>, K_PREFETCH>("A PF 32.32 wg512 fpw16 tiled occ1", true, 8);
  addA<strided_cfg<f, radix_list<16, 16, 4>, 1024, 16, 4, W>, K_PLAIN>("A 16.16.4 wg1024 fpw16 tiled", true, 8);
  addA<strided_cfg<f, radix_list<8, 16, 8>, 1024, 16, 4, W>, K_PLAIN>("A 8.16.8 wg1024 fpw16 tiled", true, 8);
  addA<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, W>, K_PLAIN>("A 16.8.8 wg1024 fpw16 tiled gpw2", true, 2);
  addA<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, W>, K_PLAIN>("A 16.8.8 wg1024 fpw16 tiled gpw8", true, 8);
  addB<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, RD>, K_PREFETCH>("B PF 16.8.8 wg1024 fpw16 tiled gpw8", true, 8);
  addB<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, RD>, K_PF_TIN>("B PF+TIN 16.8.8 wg1024 fpw16 tiled", true, 4);
  addB<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, RD>, K_PF_TIN>("B PF+TIN 16.8.8 wg1024 fpw16 tiled gpw8", true, 8);
  addB<strided_cfg<f, radix_list<32, 32>, 512, 16, 2, RD>, K_PF_TIN>("B PF+TIN 32.32 wg512 fpw16 tiled", true, 4);
  addB<strided_cfg<f, radix_list<32, 32>, 512, 16, 2, RD>, K_PF_TIN_LTW>("B PF+TIN+LTW 32.32 wg512 fpw16 tiled (modifier on loads)", true, 4);
  addB<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, RD>, K_TIN>("B 16.8.8 wg1024 fpw16 tiled TIN gpw2", true, 2);
  addB<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, RD>, K_TIN>("B 16.8.8 wg1024 fpw16 tiled TIN gpw8", true, 8);
#elif TUNE_CASE == 15
  addA<strided_cfg<f, radix_list<16, 8>, 256, 32, 2, W>, K_PLAIN>("A 16.8 wg256 fpw32 plain-layout (production)", false, 1);
  addB<strided_cfg<f, radix_list<16, 16>, 512, 32, 2, RD>, K_ROW_IN>("B 16.16 wg512 fpw32 row-staged (production)", false, 4);
  addA<strided_cfg<f, radix_list<8, 16>, 256, 16, 2, W>, K_PLAIN>("A 8.16 wg256 fpw16 tiled", true, 4);
  addA<strided_cfg<f, radix_list<16, 8>, 128, 16, 2, W>, K_PLAIN>("A 16.8 wg128 fpw16 tiled", true, 4);
  addA<strided_cfg<f, radix_list<16, 8>, 256, 16, 2, W>, K_PLAIN>("A 16.8 wg256(8pt) fpw16 tiled", true, 4);
  addA<strided_cfg<f, radix_list<16, 8>, 128, 16, 2, W>, K_PLAIN>("A 16.8 wg128 fpw16 tiled gpw8", true, 8);
  addB<strided_cfg<f, radix_list<16, 16>, 256, 16, 2, RD>, K_TIN>("B 16.16 wg256 fpw16 tiled TIN 4/CU", true, 4);
  addB<strided_cfg<f, radix_list<16, 16>, 256, 16, 2, RD>, K_TIN>("B 16.16 wg256 fpw16 tiled TIN 4/CU gpw8", true, 8);
#elif TUNE_CASE == 18
  addA<strided_cfg<f, radix_list<8, 8, 8>, 1024, 32, 2, W>, K_PLAIN>("A 8.8.8 wg1024 fpw32 plain-layout (production)", false, 2);
  addB<strided_cfg<f, radix_list<8, 8, 8>, 1024, 32, 2, RD>, K_ROW_IN>("B 8.8.8 wg1024 fpw32 row-staged (production)", false, 2);
  addA<strided_cfg<f, radix_list<8, 8, 8>, 1024, 32, 2, W>, K_PLAIN>("A 8.8.8 wg1024 fpw32 tiled", true, 2);
  addB<strided_cfg<f, radix_list<8, 8, 8>, 1024, 32, 2, RD>, K_TIN>("B 8.8.8 wg1024 fpw32 tiled TIN", true, 2);
  addB<strided_cfg<f, radix_list<8, 8, 8>, 1024, 32, 2, RD>, K_PLAIN>("B 8.8.8 wg1024 fpw32 tiled", true, 2);
  addA<strided_cfg<f, radix_list<8, 8, 8>, 1024, 32, 2, W>, K_PREFETCH>("A PF 8.8.8 wg1024 fpw32 tiled", true, 2);
  addB<strided_cfg<f, radix_list<8, 8, 8>, 1024, 32, 2, RD>, K_PREFETCH>("B PF 8.8.8 wg1024 fpw32 tiled", true, 2);
  addA<strided_cfg<f, radix_list<8, 8, 8>, 512, 16, 2, W>, K_PLAIN>("A 8.8.8 wg512 fpw16 tiled 2/CU", true, 4);
  addB<strided_cfg<f, radix_list<8, 8, 8>, 512, 16, 2, RD>, K_TIN>("B 8.8.8 wg512 fpw16 tiled TIN 2/CU", true, 4);
  addA<strided_cfg<f, radix_list<8, 8, 8>, 512, 32, 4, W>, K_HX>("A HX 8.8.8 wg512(32pt... 16pt x2) fpw32 tiled 2/CU", true, 2);
  addB<strided_cfg<f, radix_list<8, 8, 8>, 512, 32, 4, RD>, K_HX>("B HX 8.8.8 wg512 fpw32 tiled 2/CU", true, 2);
  addA<strided_cfg<f, radix_list<8, 8, 8>, 512, 16, 2, W>, K_PLAIN>("A 8.8.8 wg512 fpw16 tiled 2/CU gpw2", true, 2);
  addA<strided_cfg<f, radix_list<8, 8, 8>, 512, 16, 2, W>, K_NOSTW>("A 8.8.8 wg512 fpw16 tiled 2/CU WITHOUT stw", true, 4);
  addB<strided_cfg<f, radix_list<8, 8, 8>, 512, 16, 2, RD>, K_PF_TIN>("B PF+TIN 8.8.8 wg512 fpw16 tiled", true, 4);
  addB<strided_cfg<f, radix_list<8, 8, 8>, 512, 16, 2, RD>, K_PF_TIN_LTW>("B PF+TIN+LTW 8.8.8 wg512 fpw16 tiled (modifier on loads)", true, 4);
  addA<strided_cfg<f, radix_list<8, 8, 8>, 512, 16, 2, W>, K_PLAIN>("A 8.8.8 wg512 fpw16 tiled 2/CU gpw8", true, 8);
  addB<strided_cfg<f, radix_list<8, 8, 8>, 512, 16, 2, RD>, K_TIN>("B 8.8.8 wg512 fpw16 tiled TIN 2/CU gpw2", true, 2);
  addB<strided_cfg<f, radix_list<8, 8, 8>, 512, 16, 2, RD>, K_TIN>("B 8.8.8 wg512 fpw16 tiled TIN 2/CU gpw8", true, 8);
  addA<strided_cfg<f, radix_list<8, 8, 8>, 256, 16, 1, W>, K_PLAIN>("A 8.8.8 wg256(32pt) fpw16 tiled 2/CU", true, 4);
  addB<strided_cfg<f, radix_list<8, 8, 8>, 256, 16, 1, RD>, K_TIN>("B 8.8.8 wg256(32pt) fpw16 tiled TIN 2/CU", true, 4);
  addA<strided_cfg<f, radix_list<8, 8, 8>, 512, 16, 2, W>, K_PLAIN>("A 8.8.8 wg512 fpw16 tiled FORCED 1/CU (+16 KiB LDS)", true, 4, 16 << 10);
  addB<strided_cfg<f, radix_list<8, 8, 8>, 512, 16, 2, RD>, K_TIN>("B 8.8.8 wg512 fpw16 tiled TIN FORCED 1/CU (+16 KiB)", true, 4, 16 << 10);
  addA<sfr_cfg<f, radix_list<2, 16, 16>, 512, 32, 4, W>, K_SFR>("A SFR 2.16.16 wg512 fpw32 tiled 2/CU", true, 2);
  addB<sfr_cfg<f, radix_list<2, 16, 16>, 512, 32, 4, RD>, K_SFR>("B SFR 2.16.16 wg512 fpw32 tiled 2/CU", true, 2);
  addA<strided_cfg<f, radix_list<16, 32>, 512, 32, 2, W>, K_PLAIN>("A 16.32 wg512 fpw32 tiled", true, 2);
  addB<strided_cfg<f, radix_list<32, 16>, 512, 32, 2, RD>, K_PLAIN>("B 32.16 wg512 fpw32 tiled", true, 2);
#elif TUNE_CASE == 16
  addA<strided_cfg<f, radix_list<16, 16>, 512, 32, 2, W>, K_PLAIN>("A 16.16 wg512 fpw32 plain-layout (production)", false, 4);
  addB<strided_cfg<f, radix_list<16, 16>, 512, 32, 2, RD>, K_ROW_IN>("B 16.16 wg512 fpw32 row-staged (production)", false, 4);
  addA<strided_cfg<f, radix_list<16, 16>, 512, 32, 2, W>, K_PLAIN>("A 16.16 wg512 fpw32 tiled", true, 4);
  addB<strided_cfg<f, radix_list<16, 16>, 512, 32, 2, RD>, K_PLAIN>("B 16.16 wg512 fpw32 tiled", true, 4);
  addA<strided_cfg<f, radix_list<16, 16>, 512, 32, 2, W>, K_PREFETCH>("A PF 16.16 wg512 fpw32 tiled", true, 4);
  addB<strided_cfg<f, radix_list<16, 16>, 512, 32, 2, RD>, K_PREFETCH>("B PF 16.16 wg512 fpw32 tiled", true, 4);
  addA<strided_cfg<f, radix_list<16, 16>, 256, 16, 2, W>, K_PLAIN>("A 16.16 wg256 fpw16 tiled 4/CU", true, 4);
  addB<strided_cfg<f, radix_list<16, 16>, 256, 16, 2, RD>, K_TIN>("B 16.16 wg256 fpw16 tiled TIN 4/CU", true, 4);
  addB<strided_cfg<f, radix_list<16, 16>, 256, 16, 2, RD>, K_PLAIN>("B 16.16 wg256 fpw16 tiled 4/CU", true, 4);
  addA<strided_cfg<f, radix_list<16, 16>, 256, 16, 2, W>, K_PLAIN>("A 16.16 wg256 fpw16 tiled 4/CU gpw8", true, 8);
  addA<strided_cfg<f, radix_list<16, 16>, 256, 16, 2, W>, K_NOSTW>("A 16.16 wg256 fpw16 tiled 4/CU WITHOUT stw", true, 4);
  addB<strided_cfg<f, radix_list<16, 16>, 256, 16, 2, RD>, K_PF_TIN>("B PF+TIN 16.16 wg256 fpw16 tiled", true, 4);
  addB<strided_cfg<f, radix_list<16, 16>, 256, 16, 2, RD>, K_PF_TIN_LTW>("B PF+TIN+LTW 16.16 wg256 fpw16 tiled (modifier on loads)", true, 4);
  addB<strided_cfg<f, radix_list<16, 16>, 256, 16, 2, RD>, K_TIN>("B 16.16 wg256 fpw16 tiled TIN 4/CU gpw8", true, 8);
  addA<strided_cfg<f, radix_list<8, 8, 4>, 1024, 32, 2, W>, K_PLAIN>("A 8.8.4 wg1024 fpw32 tiled", true, 2);
  addB<strided_cfg<f, radix_list<8, 8, 4>, 1024, 32, 2, RD>, K_TIN>("B 8.8.4 wg1024 fpw32 tiled TIN", true, 2);
  addA<strided_cfg<f, radix_list<16, 16>, 512, 32, 2, W>, K_PLAIN>("A 16.16 wg512 fpw32 tiled (pairs with 8.8.4 B)", true, 4);
#elif TUNE_CASE == 22
  addA<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, W>, K_PLAIN>("A 16.16.8 wg1024 fpw8 plain-layout (production)", false, 1);
  addB<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, RD>, K_PLAIN>("B 16.16.8 wg1024 fpw8 plain (production)", false, 1);
  addA<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, W>, K_PLAIN>("A 16.16.8 wg1024 fpw8 tiled", true, 1);
  addB<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, RD>, K_TIN>("B 16.16.8 wg1024 fpw8 tiled TIN", true, 1);
  addB<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, RD>, K_PLAIN>("B 16.16.8 wg1024 fpw8 tiled", true, 1);
  addB<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, RD>, K_PF_TIN>("B PF+TIN 16.16.8 wg1024 fpw8 tiled", true, 1);
  addA<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, W>, K_PLAIN>("A 16.16.8 wg1024 fpw8 tiled gpw4", true, 4);
  addB<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, RD>, K_TIN>("B 16.16.8 wg1024 fpw8 tiled TIN gpw4", true, 4);
  addA<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, W>, K_DG>("A DG 16.16.8 wg1024 fpw8 tiled (16 columns loaded) gpw1", true, 1);
  addA<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, W>, K_DG>("A DG 16.16.8 wg1024 fpw8 tiled gpw2", true, 2);
  addA<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, W>, K_DG>("A DG 16.16.8 wg1024 fpw8 tiled gpw4", true, 4);
  addA<strided_cfg<f, radix_list<16, 16, 8>, 512, 8, 2, W>, K_DG>("A DG 16.16.8 wg512 fpw8 tiled gpw2", true, 2);
  addA<strided_cfg<f, radix_list<16, 16, 8>, 512, 8, 2, W>, K_PLAIN>("A 16.16.8 wg512 fpw8 tiled gpw4", true, 4);
  addA<strided_cfg<f, radix_list<8, 16, 16>, 1024, 8, 4, W>, K_DG>("A DG 8.16.16 wg1024 fpw8 tiled gpw2", true, 2);
  addA<sfr_cfg<f, radix_list<2, 16, 8, 8>, 1024, 16, 4, W>, K_SFR>("A SFR 2.16.8.8 wg1024 fpw16 tiled 1/CU", true, 4);
  addB<sfr_cfg<f, radix_list<2, 16, 8, 8>, 1024, 16, 4, RD>, K_SFR>("B SFR 2.16.8.8 wg1024 fpw16 tiled 1/CU", true, 4);
  addA<sfr_cfg<f, radix_list<4, 8, 8, 8>, 1024, 16, 4, W>, K_SFR>("A SFR 4.8.8.8 wg1024 fpw16 tiled", true, 4);
  addB<sfr_cfg<f, radix_list<4, 8, 8, 8>, 1024, 16, 4, RD>, K_SFR>("B SFR 4.8.8.8 wg1024 fpw16 tiled", true, 4);
  addA<strided_cfg<f, radix_list<16, 16, 8>, 1024, 16, 4, W>, K_HX>("A HX 16.16.8 wg1024(32pt) fpw16 tiled", true, 1);
  addB<strided_cfg<f, radix_list<16, 16, 8>, 1024, 16, 4, RD>, K_HX>("B HX 16.16.8 wg1024(32pt) fpw16 tiled", true, 1);
#elif TUNE_CASE == 116
  addA<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, W>, K_PLAIN>("A 16.16 wg128 fpw8 tiled (production)", true, 2);
  addB<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, RD>, K_TIN>("B 16.16 wg128 fpw8 tiled TIN (production)", true, 2);
  addB<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, RD>, K_PF_TIN>("B PF+TIN 16.16 wg128 fpw8 tiled", true, 2);
  addB<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, RD>, K_PF_TIN>("B PF+TIN 16.16 wg128 fpw8 tiled gpw4", true, 4);
  addB<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, RD>, K_PF_TIN_LTW>("B PF+TIN+LTW 16.16 wg128 fpw8 tiled (modifier on loads)", true, 2);
  addA<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, W>, K_PLAIN>("A 16.16 wg128 fpw8 tiled gpw4", true, 4);
  addA<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, W>, K_PLAIN>("A 16.16 wg128 fpw8 tiled gpw8", true, 8);
  addA<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, W>, K_PREFETCH>("A PF 16.16 wg128 fpw8 tiled", true, 4);
  addA<strided_cfg<d, radix_list<16, 16>, 256, 8, 2, W>, K_PLAIN>("A 16.16 wg256(8pt) fpw8 tiled", true, 4);
  addB<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, RD>, K_TIN>("B 16.16 wg128 fpw8 tiled TIN gpw4", true, 4);
  addB<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, RD>, K_TIN>("B 16.16 wg128 fpw8 tiled TIN gpw8", true, 8);
  addA<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, W>, K_NOSTW>("A 16.16 wg128 fpw8 tiled WITHOUT stw (timing only)", true, 2);
#elif TUNE_CASE == 118
  addA<strided_cfg<d, radix_list<8, 8, 8>, 512, 8, 2, W>, K_PLAIN>("A 8.8.8 wg512 fpw8 tiled (production)", true, 4);
  addB<strided_cfg<d, radix_list<8, 8, 8>, 512, 8, 2, RD>, K_TIN>("B 8.8.8 wg512 fpw8 tiled TIN (production)", true, 4);
  addB<strided_cfg<d, radix_list<8, 8, 8>, 512, 8, 2, RD>, K_PF_TIN>("B PF+TIN 8.8.8 wg512 fpw8 tiled", true, 4);
  addB<strided_cfg<d, radix_list<8, 8, 8>, 512, 8, 2, RD>, K_PF_TIN_LTW>("B PF+TIN+LTW 8.8.8 wg512 fpw8 tiled (modifier on loads)", true, 4);
  addA<strided_cfg<d, radix_list<8, 8, 8>, 512, 8, 2, W>, K_PREFETCH>("A PF 8.8.8 wg512 fpw8 tiled", true, 4);
  addA<strided_cfg<d, radix_list<8, 8, 8>, 256, 8, 2, W>, K_PLAIN>("A 8.8.8 wg256(16pt) fpw8 tiled", true, 4);
  addB<strided_cfg<d, radix_list<8, 8, 8>, 256, 8, 2, RD>, K_TIN>("B 8.8.8 wg256(16pt) fpw8 tiled TIN", true, 4);
  addA<strided_cfg<d, radix_list<16, 32>, 256, 8, 2, W>, K_PLAIN>("A 16.32 wg256 fpw8 tiled", true, 4);
  addB<strided_cfg<d, radix_list<32, 16>, 256, 8, 2, RD>, K_PLAIN>("B 32.16 wg256 fpw8 tiled", true, 4);
  addA<strided_cfg<d, radix_list<8, 8, 8>, 512, 8, 2, W>, K_PLAIN>("A 8.8.8 wg512 fpw8 tiled gpw8", true, 8);
  addA<strided_cfg<d, radix_list<8, 8, 8>, 512, 8, 2, W>, K_NOSTW>("A 8.8.8 wg512 fpw8 tiled WITHOUT stw (timing only)", true, 4);
#elif TUNE_CASE == 120
  addA<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, W>, K_PLAIN>("A 16.8.8 wg512 fpw8 tiled (production)", true, 4);
  addB<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, RD>, K_TIN>("B 16.8.8 wg512 fpw8 tiled TIN (production)", true, 4);
  addB<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, RD>, K_PLAIN>("B 16.8.8 wg512 fpw8 tiled", true, 4);
  addA<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, W>, K_PLAIN>("A 16.8.8 wg512 fpw8 tiled gpw2", true, 2);
  addA<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, W>, K_PLAIN>("A 16.8.8 wg512 fpw8 tiled gpw8", true, 8);
  addA<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, W>, K_PLAIN>("A 16.8.8 wg512 fpw8 tiled gpw16 (= 1x resident)", true, 16);
  addA<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, W>, K_NOSTW>("A 16.8.8 wg512 fpw8 tiled WITHOUT stw (timing only)", true, 4);
  addA<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, W>, K_PF_NOSTW>("A PF 16.8.8 wg512 fpw8 tiled WITHOUT stw", true, 4);
  addA<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, W>, K_NOSTW>("A 16.8.8 wg512 fpw8 tiled WITHOUT stw gpw8", true, 8);
  addA<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, W>, K_NOSTW>("A 16.8.8 wg512 fpw8 tiled WITHOUT stw gpw2", true, 2);
  addA<strided_cfg<d, radix_list<32, 32>, 256, 8, 1, W>, K_NOSTW>("A 32.32 wg256 fpw8 tiled WITHOUT stw", true, 4);
  addA<strided_cfg<d, radix_list<16, 8, 8>, 1024, 8, 4, W>, K_NOSTW>("A 16.8.8 wg1024(8pt) fpw8 tiled WITHOUT stw", true, 4);
  addB<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, RD>, K_PF_TIN_LTW>("B PF+TIN+LTW 16.8.8 wg512 fpw8 tiled (modifier on loads)", true, 4);
  addB<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, RD>, K_PF_TIN>("B PF+TIN 16.8.8 wg512 fpw8 tiled gpw2", true, 2);
  addB<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, RD>, K_PF_TIN>("B PF+TIN 16.8.8 wg512 fpw8 tiled gpw8", true, 8);
  addA<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, W>, K_PREFETCH>("A PF 16.8.8 wg512 fpw8 tiled", true, 4);
  addB<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, RD>, K_PREFETCH>("B PF 16.8.8 wg512 fpw8 tiled", true, 4);
  addA<sfr_cfg<d, radix_list<2, 8, 8, 8>, 512, 8, 4, W>, K_SFR>("A SFR 2.8.8.8 wg512 fpw8 tiled 2/CU", true, 4);
  addA<sfr_cfg<d, radix_list<2, 8, 8, 8>, 512, 8, 4, W>, K_SFR>("A SFR 2.8.8.8 wg512 fpw8 tiled 2/CU gpw2", true, 2);
  addB<sfr_cfg<d, radix_list<2, 8, 8, 8>, 512, 8, 4, RD>, K_SFR>("B SFR 2.8.8.8 wg512 fpw8 tiled 2/CU", true, 4);
  addB<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, RD>, K_PF_TIN>("B PF+TIN 16.8.8 wg512 fpw8 tiled", true, 4);
  addA<wg_cfg<d, radix_list<16, 8, 8>, 256, 8, 0, 0, TW_GLOBAL, 2, W, 0, 0>, K_HX>("A HX 16.8.8 wg256(32pt) fpw8 TWL0 tiled 2/CU", true, 4);
  addA<strided_cfg<d, radix_list<16, 8, 8>, 256, 8, 2, W>, K_HX>("A HX 16.8.8 wg256(32pt) fpw8 tiled 2/CU", true, 4);
  addB<strided_cfg<d, radix_list<16, 8, 8>, 256, 8, 2, RD>, K_HX>("B HX 16.8.8 wg256(32pt) fpw8 tiled 2/CU", true, 4);
#endif
  // reference result: the first A and the first B
  const int reps = 5;
  const size_t cmp_count = std::min<size_t>(total, (size_t)N * 2 * 2);  // first two transforms
  std::vector<T> ref(cmp_count);
  {
    const result r = run_pair(g_a[0], g_b[0], in, scratch, out, reps);
    CK(hipMemcpy(ref.data(), out, cmp_count * sizeof(T), hipMemcpyDeviceToHost));
    const double bytes = 2.0 * N * sizeof(cx<T>) * BATCH;
    printf("%-52s + %-50s A %7.1f us  B %7.1f us per chunk | total %.3f ms = %.3f of 8 TB/s (1x bytes)\n", g_a[0].name.c_str(), g_b[0].name.c_str(), r.a_us, r.b_us, r.total_ms, bytes / (r.total_ms * 1e-3) / 8e12);
  }
  auto partner = [&](std::vector<variant>& list, const variant& v) -> const variant* {
    for (auto& p : list) if (p.tiled == v.tiled && (!v.tiled || (p.kind != K_TIN || p.fpw == v.fpw))) return &p;
    return nullptr;
  };
  // the tiled layout's tile is stage A's group width; a stage-B variant needs n2 % t == 0 and (n2 / r0) % t == 0
  printf("---- stage-A variants (partner: first stage-B variant with the same layout)\n");
  for (size_t i = 0; i < g_a.size(); ++i) {
    const variant* pb = nullptr;
    for (auto& p : g_b) if (p.tiled == g_a[i].tiled && (N2 / p.r0) % g_a[i].fpw == 0 && ((p.kind != K_TIN && p.kind != K_PF_TIN && p.kind != K_PF_TIN_LTW) || p.fpw == g_a[i].fpw)) { pb = &p; break; }
    if (pb == nullptr) { printf("%-52s no partner\n", g_a[i].name.c_str()); continue; }
    CK(hipMemset(out, 0, cmp_count * sizeof(T)));
    const result r = run_pair(g_a[i], *pb, in, scratch, out, reps);
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, g_a[i].fn, g_a[i].wg, g_a[i].lds));
    printf("%-52s occ %d lds %6zu  A %7.1f us (%.2f TB/s)  [B %-40s %7.1f us]  rel-L2 vs ref %.1e\n", g_a[i].name.c_str(), occ, g_a[i].lds, r.a_us,
           2.0 * 268435456.0 / (r.a_us * 1e-6) * 1e-12, pb->name.c_str(), r.b_us, compare(out, ref, cmp_count));
  }
  printf("---- stage-B variants (partner: first tiled / plain stage-A variant with a matching tile)\n");
  for (size_t i = 0; i < g_b.size(); ++i) {
    const variant* pa = nullptr;
    for (auto& p : g_a) if (p.tiled == g_b[i].tiled && (!p.tiled || ((N2 / g_b[i].r0) % p.fpw == 0 && ((g_b[i].kind != K_TIN && g_b[i].kind != K_PF_TIN && g_b[i].kind != K_PF_TIN_LTW) || p.fpw == g_b[i].fpw)))) { pa = &p; break; }
    if (pa == nullptr) { printf("%-52s no partner\n", g_b[i].name.c_str()); continue; }
    CK(hipMemset(out, 0, cmp_count * sizeof(T)));
    const result r = run_pair(*pa, g_b[i], in, scratch, out, reps);
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, g_b[i].fn, g_b[i].wg, g_b[i].lds));
    printf("%-52s occ %d lds %6zu  B %7.1f us (%.2f TB/s)  [A %-40s %7.1f us]  rel-L2 vs ref %.1e\n", g_b[i].name.c_str(), occ, g_b[i].lds, r.b_us,
           2.0 * 268435456.0 / (r.b_us * 1e-6) * 1e-12, pa->name.c_str(), r.a_us, compare(out, ref, cmp_count));
  }
  (void)partner;
  // the modifier moved from stage A's stores to stage B's loads: the pair's result against the reference
  {
    const variant *pa = nullptr, *pb = nullptr;
    for (auto& p : g_a) if (p.kind == K_NOSTW && p.tiled) { pa = &p; break; }
    for (auto& p : g_b) if (p.kind == K_PF_TIN_LTW && pa != nullptr && p.fpw == pa->fpw) { pb = &p; break; }
    if (pa != nullptr && pb != nullptr) {
      CK(hipMemset(out, 0, cmp_count * sizeof(T)));
      const result r = run_pair(*pa, *pb, in, scratch, out, reps);
      printf("---- modifier on stage B's loads: %s + %s: A %.1f us  B %.1f us per chunk, rel-L2 vs ref %.1e\n", pa->name.c_str(), pb->name.c_str(),
             r.a_us, r.b_us, compare(out, ref, cmp_count));
    }
  }
  return 0;
}
