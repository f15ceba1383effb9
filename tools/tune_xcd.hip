// Tuner / parity check of the XCD-local four-step kernel (portfft_amd/csrc/stockham_xcd.hpp) against the two-launch
// production pair of the same stage kernels (Infinity-Cache-sized chunks, writer / reader policies), bit for bit.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTUNE_CASE=<case> tools/tune_xcd.hip -o build/tune_xcd_<case>
//   cases: 16 = fp32 65536 (256 x 256), 17 = fp32 2^17 (256 x 512), 18 = fp32 2^18 (512 x 512), 116 / 117 / 118 = fp64 65536 /
//          2^17 / 2^18, 120 = fp64 2^20 (1024 x 1024, BASELINE config 3); -DTUNE_WG=512: work-groups of 512 lanes (groups side by side)
//   env:   TUNE_BATCH, TUNE_SWEEP=1 (slots / lag / work-groups per CU sweep), TUNE_SLOTS, TUNE_LAG, TUNE_LOOKAHEAD,
//          TUNE_WG_PER_CU, TUNE_REPS
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#ifndef TUNE_NO_PROF
#define PFA_XCD_PROF 1
#endif
#include "../portfft_amd/csrc/stockham_xcd.hpp"
#include "../portfft_amd/csrc/kernels.hpp"
#ifdef TUNE_GANG
#include "probes/xcd_gang.hpp"
#endif
using namespace pfa;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

#ifndef TUNE_CASE
#define TUNE_CASE 16
#endif
#ifndef TUNE_OCCX
#define TUNE_OCCX 2  // waves per SIMD the fused kernel's register budget leaves room for
#endif
#ifndef TUNE_FREERUN
#define TUNE_FREERUN 0  // 1: timing experiment without claims and hand-off waits (results are garbage)
#endif
#ifndef TUNE_WG
#define TUNE_WG 0  // lanes of the fused launch's work-groups (0: the larger of the two configurations')
#endif
#if TUNE_CASE == 116
using T = double;
using CfgA = strided_cfg<double, radix_list<16, 16>, 128, 8, 2, PFA_AUX_NT>;
using CfgB = CfgA;
constexpr long long DEF_BATCH = 2048;
#elif TUNE_CASE == 118
using T = double;
using CfgA = strided_cfg<double, radix_list<8, 8, 8>, 512, 8, 2, PFA_AUX_NT>;
using CfgB = CfgA;
constexpr long long DEF_BATCH = 256;
#elif TUNE_CASE == 120
using T = double;
using CfgA = strided_cfg<double, radix_list<16, 8, 8>, 512, 8, 2, PFA_AUX_NT>;
using CfgB = CfgA;
constexpr long long DEF_BATCH = 128;
#elif TUNE_CASE == 18
using T = float;
using CfgA = strided_cfg<float, radix_list<8, 8, 8>, 512, 16, 2, PFA_AUX_NT>;
using CfgB = CfgA;
constexpr long long DEF_BATCH = 1024;
#elif TUNE_CASE == 17  // fp32 2^17 = 256 x 512: two 256-lane stage-A groups or one 512-lane stage-B group per task
using T = float;
using CfgA = strided_cfg<float, radix_list<16, 16>, 256, 16, 2, PFA_AUX_NT>;
using CfgB = strided_cfg<float, radix_list<8, 8, 8>, 512, 16, 2, PFA_AUX_NT>;
constexpr long long DEF_BATCH = 2048;
#elif TUNE_CASE == 19  // fp32 2^19 = 512 x 1024
using T = float;
using CfgA = strided_cfg<float, radix_list<8, 8, 8>, 512, 16, 2, PFA_AUX_NT>;
using CfgB = strided_cfg<float, radix_list<32, 32>, 512, 16, 2, PFA_AUX_NT>;
constexpr long long DEF_BATCH = 512;
#elif TUNE_CASE == 20  // fp32 2^20 = 1024 x 1024
using T = float;
using CfgA = strided_cfg<float, radix_list<16, 8, 8>, 1024, 16, 4, PFA_AUX_NT>;
using CfgB = CfgA;
constexpr long long DEF_BATCH = 256;
#elif TUNE_CASE == 15  // fp32 2^15 = 128 x 256
using T = float;
using CfgA = strided_cfg<float, radix_list<16, 8>, 128, 16, 2, PFA_AUX_NT>;
using CfgB = strided_cfg<float, radix_list<16, 16>, 256, 16, 2, PFA_AUX_NT>;
constexpr long long DEF_BATCH = 8192;
#elif TUNE_CASE == 191  // fp32 2^19 = 1024 x 512: two 512-lane stage-B groups side by side
using T = float;
using CfgA = strided_cfg<float, radix_list<16, 8, 8>, 1024, 16, 4, PFA_AUX_NT>;
using CfgB = strided_cfg<float, radix_list<8, 8, 8>, 512, 16, 2, PFA_AUX_NT>;
constexpr long long DEF_BATCH = 512;
#elif TUNE_CASE == 119  // fp64 2^19 = 512 x 1024
using T = double;
using CfgA = strided_cfg<double, radix_list<8, 8, 8>, 512, 8, 2, PFA_AUX_NT>;
using CfgB = strided_cfg<double, radix_list<16, 8, 8>, 512, 8, 2, PFA_AUX_NT>;
constexpr long long DEF_BATCH = 256;
#elif TUNE_CASE == 121  // fp64 2^20 = 1024 x 1024 on 1024 lanes (16 waves per CU instead of 8): four passes
using T = double;
using CfgA = strided_cfg<double, radix_list<8, 8, 4, 4>, 1024, 8, 4, PFA_AUX_NT>;
using CfgB = CfgA;
constexpr long long DEF_BATCH = 128;
#elif TUNE_CASE == 122  // the same, radices 4.4.8.8
using T = double;
using CfgA = strided_cfg<double, radix_list<4, 4, 8, 8>, 1024, 8, 4, PFA_AUX_NT>;
using CfgB = CfgA;
constexpr long long DEF_BATCH = 128;
#elif TUNE_CASE == 117  // fp64 2^17 = 256 x 512
using T = double;
using CfgA = strided_cfg<double, radix_list<16, 16>, 128, 8, 2, PFA_AUX_NT>;
using CfgB = strided_cfg<double, radix_list<8, 8, 8>, 512, 8, 2, PFA_AUX_NT>;
constexpr long long DEF_BATCH = 1024;
#else
using T = float;
using CfgA = strided_cfg<float, radix_list<16, 16>, 256, 16, 2, PFA_AUX_NT>;
using CfgB = CfgA;
constexpr long long DEF_BATCH = 4096;
#endif
constexpr int XWG = TUNE_WG != 0 ? TUNE_WG : (CfgA::WG > CfgB::WG ? CfgA::WG : CfgB::WG);
using XL = xcd_layout<CfgA, CfgB, XWG>;
constexpr long long N1 = CfgA::N, N2 = CfgB::N, N = N1 * N2;
using CfgW = typename xcd_with_aux<CfgA, PFA_AUX_WRITER>::type;
using CfgR = typename xcd_with_aux<CfgB, PFA_AUX_READER>::type;

static long long g_batch = DEF_BATCH;
static int g_cus = 256;

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
__global__ void count_diff(const unsigned* a, const unsigned* b, size_t n, unsigned long long* out) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}
__global__ void xcc_census(unsigned* o) {
  if (threadIdx.x == 0) { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); atomicMax(o, (v & 0xf) + 1); }
}

static int g_stw_levels, g_stw_shift;
static void* g_stw_tab;
static cx<T>*g_tw_a, *g_tw_b;

static strided_args args_a(const T* in, T* scratch, long long nb, bool slots) {
  strided_args a{};
  const int t = CfgA::FPW;
  a.in = in; a.out = scratch; a.tw = g_tw_a; a.total = nb * N2; a.inner = N2;
  if (slots) { a.twl_lds_off = (unsigned)(XL::TWL_A * sizeof(cx<T>)); a.stw_lds_off = (unsigned)(XL::STW * sizeof(cx<T>)); }
  a.in_dist_outer = N; a.out_dist_outer = slots ? 0 : N; a.in_stride = (unsigned)N2; a.in_fdist = 1;
  a.scale = 1.0; a.stw_tab = g_stw_tab; a.stw_levels = g_stw_levels; a.stw_lshift = g_stw_shift; a.stw_cdiv = 1;
  a.out_gdist = N1 * t; a.out_stride = (unsigned)t; a.out_fdist = 1;
  return a;
}
static strided_args args_b(const T* scratch, T* out, long long nb, bool slots) {
  strided_args a{};
  const int t = CfgA::FPW;
  a.in = scratch; a.out = out; a.tw = g_tw_b; a.total = nb * N1; a.inner = N1;
  if (slots) { a.twl_lds_off = (unsigned)(XL::TWL_B * sizeof(cx<T>)); a.stw_lds_off = (unsigned)(XL::STW * sizeof(cx<T>)); }
  a.in_dist_outer = slots ? 0 : N; a.out_dist_outer = N; a.out_stride = (unsigned)N1; a.out_fdist = 1;
  a.scale = 1.0; a.stw_cdiv = 1;
  int sh = 0; while ((1 << sh) < t) ++sh;
  a.in_tile_shift = sh; a.in_stride = (unsigned)(N1 * t); a.in_fdist = (unsigned)t;
  return a;
}

static unsigned grid_of(const void* fn, int wg, size_t lds, long long groups, int gpw) {
  int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, wg, lds));
  per_cu = std::max(per_cu, 1);
  const long long resident = (long long)per_cu * g_cus;
  long long grid = (groups + gpw - 1) / gpw;
  grid = std::min(groups, std::max(grid, std::min<long long>(groups, 2 * resident)));
  return (unsigned)std::max<long long>(1, grid);
}
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); g_cus = prop.multiProcessorCount;
  if (const char* e = getenv("TUNE_BATCH")) g_batch = atoll(e);
  const int reps = getenv("TUNE_REPS") ? atoi(getenv("TUNE_REPS")) : 7;
  const size_t total = (size_t)g_batch * N * 2;  // scalars
  T *in, *out, *ref, *scratch;
  CK(hipMalloc(&in, total * sizeof(T))); CK(hipMalloc(&out, total * sizeof(T))); CK(hipMalloc(&ref, total * sizeof(T)));
  CK(hipMalloc(&scratch, (size_t)2048 << 20));
  fill_uniform<<<4096, 256>>>(in, total, 7);
  CK(hipDeviceSynchronize());
  {
    int bits = 0; while ((1ll << bits) < N) ++bits;
    for (int l = 1; l <= 4; ++l) {
      const int sh = (bits + l - 1) / l;
      if (((size_t)l << sh) * sizeof(cx<T>) <= 16 * 1024) { g_stw_levels = l; g_stw_shift = sh; break; }
    }
    const long long per = 1ll << g_stw_shift;
    std::vector<cx<T>> tab((size_t)g_stw_levels * per);
    for (int l = 0; l < g_stw_levels; ++l) for (long long i = 0; i < per; ++i) {
      const long long m = (long long)(((unsigned long long)i << (l * g_stw_shift)) % (unsigned long long)N);
      const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)N;
      tab[l * per + i] = {(T)cosl(a), (T)sinl(a)};
    }
    CK(hipMalloc(&g_stw_tab, tab.size() * sizeof(tab[0]))); CK(hipMemcpy(g_stw_tab, tab.data(), tab.size() * sizeof(tab[0]), hipMemcpyHostToDevice));
  }
  g_tw_a = make_twiddles<typename CfgA::Seq>();
  g_tw_b = make_twiddles<typename CfgB::Seq>();
  const size_t stw_bytes = ((size_t)g_stw_levels << g_stw_shift) * sizeof(cx<T>);
  const double bytes = 2.0 * N * sizeof(cx<T>) * g_batch;
  printf("N = %lld x %lld, batch %lld, %s, %d CUs, transform %zu KiB\n", N1, N2, g_batch, sizeof(T) == 4 ? "fp32" : "fp64", g_cus, (size_t)(N * sizeof(cx<T>)) >> 10);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // ---- reference: the production pair, two launches per 256 MiB chunk
  {
    const void* fa = (const void*)&stockham_strided_kernel<CfgW, false, 1>;
    const void* fb = (const void*)&stockham_strided_kernel<CfgR, false, 0, 0, 1>;
    const size_t lds_a = strided_lds_bytes<CfgA>() + stw_bytes, lds_b = strided_lds_bytes<CfgB>();
    CK(hipFuncSetAttribute(fa, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
    CK(hipFuncSetAttribute(fb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    const size_t per = (size_t)N * sizeof(cx<T>);
    const long long chunk = std::max<long long>(1, std::min<long long>(g_batch, (long long)(((size_t)256 << 20) / per)));
    std::vector<double> tt;
    for (int rep = 0; rep <= reps; ++rep) {
      CK(hipEventRecord(e0));
      for (long long b0 = 0; b0 < g_batch; b0 += chunk) {
        const long long nb = std::min(chunk, g_batch - b0);
        const strided_args aa = args_a(in + 2 * b0 * N, scratch, nb, false);
        hipLaunchKernelGGL((stockham_strided_kernel<CfgW, false, 1>), dim3(grid_of(fa, CfgA::WG, lds_a, nb * N2 / CfgA::FPW, 4)), dim3(CfgA::WG), lds_a, 0, aa);
        const strided_args ab = args_b(scratch, ref + 2 * b0 * N, nb, false);
        hipLaunchKernelGGL((stockham_strided_kernel<CfgR, false, 0, 0, 1>), dim3(grid_of(fb, CfgB::WG, lds_b, nb * N1 / CfgB::FPW, 4)), dim3(CfgB::WG), lds_b, 0, ab);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) tt.push_back(ms);
    }
    CK(hipGetLastError());
    printf("two launches per 256 MiB chunk (production pair):  %.3f ms = %.3f of 8 TB/s\n", median(tt), bytes / (median(tt) * 1e-3) / 8e12);
  }
  // ---- the XCD-local launch
  unsigned* d_census; CK(hipMalloc(&d_census, 4)); CK(hipMemset(d_census, 0, 4));
  xcc_census<<<4096, 64>>>(d_census);
  unsigned n_queues = 0; CK(hipMemcpy(&n_queues, d_census, 4, hipMemcpyDeviceToHost));
  printf("XCC ids seen: %u\n", n_queues);
  unsigned long long* d_diff; CK(hipMalloc(&d_diff, 8));
  const void* fx = (const void*)&stockham_xcd_fourstep_kernel<CfgA, CfgB, false, 1, 1, TUNE_FREERUN, TUNE_OCCX, XWG>;
  hipFuncAttributes fattr; CK(hipFuncGetAttributes(&fattr, fx));
  printf("fused kernel: %d VGPRs, %d SGPRs... numRegs %d, static LDS %zu\n", fattr.numRegs, 0, fattr.numRegs, fattr.sharedSizeBytes);
  auto run = [&](int slots, int lag, int lookahead, int wg_per_cu, bool verbose) {
    const int map_log2 = 8;
    const size_t own = xcd_lds_bytes<CfgA, CfgB, XWG>(stw_bytes);
    // pad the LDS request so that exactly wg_per_cu work-groups fit a CU
    size_t lds = own;
    if (wg_per_cu > 0) {
      const size_t cu = 160 * 1024;
      const size_t want = (cu / wg_per_cu) & ~(size_t)15;
      if (want >= own) lds = std::max(own, std::min(want, (cu / (wg_per_cu + 1) + 16 + 15) & ~(size_t)15));
    }
    CK(hipFuncSetAttribute(fx, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fx, XWG, lds));
    const unsigned grid = (unsigned)(per_cu * g_cus);
    const unsigned words = xcd_ctl_words((int)n_queues, slots, map_log2);
    unsigned* ctl; CK(hipMalloc(&ctl, words * 4)); CK(hipMemset(ctl, 0, words * 4));
    const size_t scratch_need = (size_t)n_queues * (size_t)slots * N * sizeof(cx<T>);
    if (scratch_need > ((size_t)2048 << 20)) { printf("scratch too small\n"); return; }
    xcd_args x{};
    x.a = args_a(in, scratch, g_batch, true);
    x.b = args_b(scratch, out, g_batch, true);
    x.ctl = ctl; x.batch = g_batch; x.n_queues = (int)n_queues; x.slots = slots; x.map_log2 = map_log2;
    x.lag = lag; x.lookahead = lookahead; x.max_iters = (unsigned)((g_batch + lag + lookahead + 6) * (N1 / CfgA::FPW + N2 / CfgA::FPW)); x.lds_ctl_off = (unsigned)(own - XCD_LDS_CTL_BYTES);
    unsigned long long* d_prof; CK(hipMalloc(&d_prof, 128)); x.prof = d_prof;
    std::vector<double> tt;
    unsigned long long bad = 0; unsigned tmo = 0;
    for (int rep = 0; rep <= reps; ++rep) {
      if (rep == 0) CK(hipMemset(out, 0xff, total * sizeof(T)));
      CK(hipMemset(d_prof, 0, 128));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((stockham_xcd_fourstep_kernel<CfgA, CfgB, false, 1, 1, TUNE_FREERUN, TUNE_OCCX, XWG>), dim3(grid), dim3(XWG), lds, 0, x);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) tt.push_back(ms);
      if (rep == 0) printf("   first launch %.3f ms\n", ms);
      if (rep == 0 || rep == reps) {
        CK(hipMemset(d_diff, 0, 8));
        count_diff<<<4096, 256>>>((const unsigned*)out, (const unsigned*)ref, total * sizeof(T) / 4, d_diff);
        unsigned long long b; CK(hipMemcpy(&b, d_diff, 8, hipMemcpyDeviceToHost)); bad += b;
        std::vector<unsigned> h(words); CK(hipMemcpy(h.data(), ctl, words * 4, hipMemcpyDeviceToHost));
        tmo += h[XCD_W_TIMEOUT];
        if (h[XCD_W_TIMEOUT]) printf("   rep %d: timeouts %u, first: site %u k %u want %u saw %u after %u polls\n", rep, h[XCD_W_TIMEOUT], h[XCD_W_TIMEOUT + 1],
                                     h[XCD_W_TIMEOUT + 2], h[XCD_W_TIMEOUT + 3], h[XCD_W_TIMEOUT + 4], h[XCD_W_TIMEOUT + 5]);
        CK(hipMemset(ctl + XCD_W_TIMEOUT, 0, 32));
        unsigned dirty = 0;
        const unsigned qw = xcd_queue_words(slots, map_log2), mw = 4u << map_log2;
        for (unsigned i = 0; i < words; ++i) {
          const bool map_word = i >= XCD_W_QUEUES && (i - XCD_W_QUEUES) % qw >= 32u && (i - XCD_W_QUEUES) % qw < 32u + mw;
          if (!map_word && i != XCD_W_EPOCH && (i < XCD_W_TIMEOUT || i >= XCD_W_QUEUES) && h[i] != 0) ++dirty;
        }
        if (dirty) printf("   !! control block not clean after the launch: %u words\n", dirty);
      }
    }
    CK(hipGetLastError());
    const double ms = median(tt);
    printf("XCD-local  S=%2d lag=%d look=%d  %d WG/CU (grid %4u, lds %6zu)  %.3f ms = %.3f of 8 TB/s  min %.3f  %s\n", slots, lag, lookahead,
           per_cu, grid, lds, ms, bytes / (ms * 1e-3) / 8e12, *std::min_element(tt.begin(), tt.end()),
           (bad || tmo) ? "!! MISMATCH / TIMEOUT" : "bit-identical");
    if (bad || tmo) printf("   !! %llu mismatching words, timeouts %u\n", bad, tmo);
#ifdef PFA_XCD_PROF
    if (verbose) {
      unsigned long long h[16]; CK(hipMemcpy(h, d_prof, 128, hipMemcpyDeviceToHost));
      const double tot = (double)h[0], na = (double)h[8], nb = (double)h[11];
      printf("   wave 0, per task (us): A: pass0 %.2f middle %.2f slot wait %.2f last pass %.2f store drain %.2f | B: input wait %.2f pass0 %.2f rest %.2f | claim wait per iteration %.2f; null iterations %llu; lifetime per WG %.1f us; waits %.1f %% of it\n",
             h[4] / na / 100, h[5] / na / 100, h[2] / na / 100, h[6] / na / 100, h[7] / na / 100, h[3] / nb / 100, h[9] / nb / 100, h[10] / nb / 100,
             h[1] / (na + nb + h[12]) / 100, h[12], tot / grid / 100, 100.0 * (h[1] + h[2] + h[3]) / tot);
    }
#endif
    CK(hipFree(d_prof));
    (void)verbose;
    CK(hipFree(ctl));
  };
#ifdef TUNE_GANG
  // ---- prototype: gang-synchronous launch (tools/probes/xcd_gang.hpp); slots per gang = TUNE_GANG (1 or 2)
  auto run_gang = [&](int wg_per_cu, int prefetch) {
    constexpr int GS = TUNE_GANG;
    const void* fg = (const void*)&gang_fourstep_kernel<CfgA, CfgB, false, 1, 1, TUNE_OCCX, XWG, GS>;
    const size_t own = xcd_lds_bytes<CfgA, CfgB, XWG>(stw_bytes);
    size_t lds = own;
    const size_t cu = 160 * 1024;
    const size_t want = (cu / wg_per_cu) & ~(size_t)15;
    if (want >= own) lds = std::max(own, std::min(want, (cu / (wg_per_cu + 1) + 16 + 15) & ~(size_t)15));
    CK(hipFuncSetAttribute(fg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fg, XWG, lds));
    const unsigned grid = (unsigned)(per_cu * g_cus);
    const size_t ctl_words = 4096 + (size_t)n_queues * GANG_MAXG * 64;
    unsigned* gctl; CK(hipMalloc(&gctl, ctl_words * 4));
    xcd_args x{};
    x.a = args_a(in, scratch, g_batch, true);
    x.b = args_b(scratch, out, g_batch, true);
    x.batch = g_batch; x.n_queues = (int)n_queues; x.lds_ctl_off = (unsigned)(own - XCD_LDS_CTL_BYTES);
    std::vector<double> tt;
    unsigned long long bad = 0; unsigned tmo = 0;
    for (int rep = 0; rep <= reps; ++rep) {
      if (rep == 0) CK(hipMemset(out, 0xff, total * sizeof(T)));
      CK(hipMemset(gctl, 0, ctl_words * 4));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((gang_fourstep_kernel<CfgA, CfgB, false, 1, 1, TUNE_OCCX, XWG, GS>), dim3(grid), dim3(XWG), lds, 0, x, gctl, prefetch);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) tt.push_back(ms);
      if (rep == 0 || rep == reps) {
        CK(hipMemset(d_diff, 0, 8));
        count_diff<<<4096, 256>>>((const unsigned*)out, (const unsigned*)ref, total * sizeof(T) / 4, d_diff);
        unsigned long long b; CK(hipMemcpy(&b, d_diff, 8, hipMemcpyDeviceToHost)); bad += b;
        unsigned h[3]; CK(hipMemcpy(h, gctl, 12, hipMemcpyDeviceToHost)); tmo += h[2];
      }
    }
    CK(hipGetLastError());
    const double ms = median(tt);
    printf("gang  slots %d  prefetch %d  %d WG/CU (grid %4u, lds %6zu, ring %zu KiB per XCD)  %.3f ms = %.3f of 8 TB/s  min %.3f  %s\n", GS, prefetch, per_cu, grid, lds,
           (size_t)(grid / n_queues / (N2 / CfgA::FPW)) * GS * (size_t)(N * sizeof(cx<T>)) >> 10, ms, bytes / (ms * 1e-3) / 8e12,
           *std::min_element(tt.begin(), tt.end()), (bad || tmo) ? "!! MISMATCH / TIMEOUT" : "bit-identical");
    if (bad || tmo) printf("   !! %llu mismatching words, spins that gave up %u\n", bad, tmo);
    CK(hipFree(gctl));
  };
  for (int w : {1, 2, 3, 4}) for (int pf : {0, 1}) run_gang(w, pf);
#endif
  const int slots_log2 = getenv("TUNE_SLOTS") ? atoi(getenv("TUNE_SLOTS")) : 8;
  const int lag = getenv("TUNE_LAG") ? atoi(getenv("TUNE_LAG")) : 3;
  const int look = getenv("TUNE_LOOKAHEAD") ? atoi(getenv("TUNE_LOOKAHEAD")) : 4;
  const int wpc = getenv("TUNE_WG_PER_CU") ? atoi(getenv("TUNE_WG_PER_CU")) : 0;
  run(slots_log2, lag, look, wpc, true);
  if (getenv("TUNE_SWEEP")) {
    const int pts[][2] = {{3, 2}, {4, 2}, {4, 3}, {5, 3}, {5, 4}, {6, 4}, {6, 5}, {8, 5}, {8, 7}, {12, 8}, {16, 8}, {24, 12}, {32, 16}, {32, 24}, {48, 24}, {64, 32}};
    for (int w : {2, 3, 4}) for (auto& pt : pts) run(pt[0], pt[1], look, w, true);  // slots > lag
  }
  return 0;
}
