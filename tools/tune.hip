// Scratch tuner: interleaved timing of work-group kernel variants for one length (compile with -DTUNE_CASE=n).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include "../portfft_amd/csrc/stockham_xlane.hpp"
#include "../portfft_amd/csrc/stockham_wg_hx.hpp"
using namespace pfa;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <typename Seq, typename T>
std::vector<cx<T>> make_twiddles() {
  std::vector<cx<T>> tw(Seq::tw_total > 0 ? Seq::tw_total : 1);
  for (int p = 1; p < Seq::count; ++p) {
    const int R = Seq::r[p], Ns = Seq::ns(p);
    for (int t = 1; t < R; ++t) for (int q = 0; q < Ns; ++q) {
      const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)(t * q) / (long double)(Ns * R);
      tw[Seq::tw_off(p) + (t - 1) * Ns + q] = {(T)cosl(a), (T)sinl(a)};
    }
  }
  return tw;
}

struct variant { std::string name; int fpw; int wg; size_t lds; const void* fn; std::function<void(unsigned, long long)> launch; };
static std::vector<variant> g_variants;
static void *g_in, *g_out;

template <typename Cfg>
void add_xlane(const char* name) {
  using T = typename Cfg::T;
  auto tw = make_twiddles<typename Cfg::Seq, T>();
  cx<T>* d_tw;
  CK(hipMalloc(&d_tw, tw.size() * sizeof(cx<T>)));
  CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  const void* fn = (const void*)&stockham_wg_xlane_kernel<Cfg, false>;
  CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES));
  g_variants.push_back({name, Cfg::FPW, Cfg::WG, Cfg::LDS_BYTES, fn, [d_tw](unsigned grid, long long nfft) {
    hipLaunchKernelGGL((stockham_wg_xlane_kernel<Cfg, false>), dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, 0, (const cx<T>*)g_in, (cx<T>*)g_out, d_tw, nfft, (T)1);
  }});
}

/// register-resident form (stockham_wg_hx.hpp)
template <typename Cfg>
void add_hx(const char* name) {
  using T = typename Cfg::T;
  auto tw = make_twiddles<typename Cfg::Seq, T>();
  cx<T>* d_tw;
  CK(hipMalloc(&d_tw, tw.size() * sizeof(cx<T>)));
  CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  const void* fn = (const void*)&stockham_wg_hx_kernel<Cfg, false>;
  constexpr size_t lds = wg_hx_lds_bytes<Cfg>();
  if (lds > 160 * 1024) { printf("%s: %zu bytes of LDS -- skipped\n", name, lds); return; }
  CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  g_variants.push_back({name, 1, Cfg::WG, lds, fn, [d_tw](unsigned grid, long long nfft) {
    hipLaunchKernelGGL((stockham_wg_hx_kernel<Cfg, false>), dim3(grid), dim3(Cfg::WG), wg_hx_lds_bytes<Cfg>(), 0, (const cx<T>*)g_in, (cx<T>*)g_out, d_tw, nfft, (T)1);
  }});
}

/// ... software-pipelined (PF = 1: the next transform's loads in flight behind the first exchange) or with the next
/// transform's first half travelling by LDS-DMA into the idle image behind the last exchange (PF = 2)
template <typename Cfg, int PF = 1>
void add_hx_pf(const char* name) {
  using T = typename Cfg::T;
  auto tw = make_twiddles<typename Cfg::Seq, T>();
  cx<T>* d_tw;
  CK(hipMalloc(&d_tw, tw.size() * sizeof(cx<T>)));
  CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  const void* fn = (const void*)&stockham_wg_hx_kernel<Cfg, false, PF>;
  constexpr size_t lds = wg_hx_lds_bytes<Cfg>();
  if (lds > 160 * 1024) { printf("%s: %zu bytes of LDS -- skipped\n", name, lds); return; }
  CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  g_variants.push_back({name, 1, Cfg::WG, lds, fn, [d_tw](unsigned grid, long long nfft) {
    hipLaunchKernelGGL((stockham_wg_hx_kernel<Cfg, false, PF>), dim3(grid), dim3(Cfg::WG), wg_hx_lds_bytes<Cfg>(), 0, (const cx<T>*)g_in, (cx<T>*)g_out, d_tw, nfft, (T)1);
  }});
}

template <typename Cfg, bool PF>
void add(const char* name) {
  using T = typename Cfg::T;
  auto tw = make_twiddles<typename Cfg::Seq, T>();
  cx<T>* d_tw;
  CK(hipMalloc(&d_tw, tw.size() * sizeof(cx<T>)));
  CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  const void* fn;
  if constexpr (PF) fn = (const void*)&stockham_wg_prefetch_kernel<Cfg, false>; else fn = (const void*)&stockham_wg_kernel<Cfg, false>;
  CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES));
  g_variants.push_back({name, Cfg::FPW, Cfg::WG, Cfg::LDS_BYTES, fn, [d_tw](unsigned grid, long long nfft) {
    if constexpr (PF) hipLaunchKernelGGL((stockham_wg_prefetch_kernel<Cfg, false>), dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, 0, (const cx<T>*)g_in, (cx<T>*)g_out, d_tw, nfft, (T)1, 0ll, 4);
    else hipLaunchKernelGGL((stockham_wg_kernel<Cfg, false>), dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, 0, (const cx<T>*)g_in, (cx<T>*)g_out, d_tw, nfft, (T)1);
  }});
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  CK(hipMalloc(&g_in, bytes)); CK(hipMalloc(&g_out, bytes));
  {  // random input (a constant fill makes on-die paths look faster than they are: profiles/r2_notes.md)
    std::vector<unsigned> h((size_t)1 << 22);
    unsigned sd = 12345u;
    for (auto& w : h) { sd = sd * 1664525u + 1013904223u; w = 0x3c000000u | (sd >> 9); }  // floats in [0.0078, 0.031)
    for (size_t off = 0; off < bytes; off += h.size() * 4) CK(hipMemcpy((char*)g_in + off, h.data(), std::min(h.size() * 4, bytes - off), hipMemcpyHostToDevice));
  }
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  using f = float; using d = double;
  constexpr int NT = 2;
#if TUNE_CASE == 2048
  using S = radix_list<16, 16, 8>; using T = f; const int N = 2048;
  add<wg_cfg<f, S, 256, 2, 16, 1, TW_GLOBAL, 4, NT>, false>("twG fpw2 o4");
  add<wg_cfg<f, S, 256, 2, 16, 1, TW_REGS, 4, NT>, false>("twR fpw2 o4");
  add<wg_cfg<f, S, 256, 2, 16, 1, TW_GLOBAL, 4, NT, 0, 1>, false>("twG fpw2 o4 TWL1");
  add<wg_cfg<f, S, 256, 2, 16, 1, TW_GLOBAL, 4, NT, 0, 2>, false>("twG fpw2 o4 TWL2");
  add<wg_cfg<f, S, 256, 2, 16, 1, TW_REGS, 3, NT>, true>("twR fpw2 o3 PF");
  add<wg_cfg<f, S, 256, 2, 16, 1, TW_GLOBAL, 4, NT>, true>("twG fpw2 o4 PF");
  add<wg_cfg<f, S, 128, 1, 16, 1, TW_REGS, 4, NT>, false>("twR wg128 fpw1 o4");
  add<wg_cfg<f, S, 128, 1, 16, 1, TW_REGS, 3, NT>, true>("twR wg128 fpw1 o3 PF");
#elif TUNE_CASE == 1024
  using S = radix_list<16, 8, 8>; using T = f; const int N = 1024;
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 4, NT>, false>("twG fpw4 o4");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_REGS, 4, NT>, false>("twR fpw4 o4");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_REGS, 3, NT>, true>("twR fpw4 o3 PF");
  add<wg_cfg<f, radix_list<32, 32>, 256, 8, 32, 1, TW_REGS, 2, NT>, false>("r32x32 twR fpw8 o2");
  add<wg_cfg<f, radix_list<32, 32>, 256, 8, 32, 1, TW_GLOBAL, 2, NT>, false>("r32x32 twG fpw8 o2");
  add<wg_cfg<f, S, 64, 1, 16, 1, TW_REGS, 4, NT>, false>("twR wg64 fpw1 o4");
#elif TUNE_CASE == 8192
  using S = radix_list<32, 16, 16>; using T = f; const int N = 8192;
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 2, NT>, false>("r32.16.16 twG wg256 o2");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 2, NT>, false>("r32.16.16 twR wg256 o2");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>, false>("r32.16.16 twG TWL1 o2");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 3, NT, 0, 1>, false>("r32.16.16 twG TWL1 o3");
  add<wg_cfg<f, radix_list<16, 16, 32>, 256, 1, 16, 1, TW_GLOBAL, 2, NT>, false>("r16.16.32 twG wg256 o2");
  add<wg_cfg<f, radix_list<16, 16, 16, 2>, 512, 1, 16, 1, TW_GLOBAL, 4, NT>, false>("r16.16.16.2 twG wg512 o4");
  add<wg_cfg<f, radix_list<8, 8, 8, 16>, 512, 1, 16, 1, TW_GLOBAL, 4, NT>, false>("r8.8.8.16 twG wg512 o4");
  add<wg_cfg<f, radix_list<16, 8, 8, 8>, 512, 1, 16, 1, TW_GLOBAL, 4, NT>, false>("r16.8.8.8 twG wg512 o4");
#elif TUNE_CASE == 32768  // beyond LDS: register-resident forms
  using T = f; const int N = 32768;
  add_hx<wg_cfg<f, radix_list<32, 32, 32>, 512, 1, 32, 1, TW_GLOBAL, 2, NT, 0, 1>>("hx 32.32.32 wg512 TWL1");
  add_hx<wg_cfg<f, radix_list<32, 32, 32>, 512, 1, 32, 1, TW_GLOBAL, 2, NT, 0, 0>>("hx 32.32.32 wg512 TWL0");
  add_hx<wg_cfg<f, radix_list<32, 32, 32>, 1024, 1, 32, 1, TW_GLOBAL, 4, NT, 0, 1>>("hx 32.32.32 wg1024 TWL1");
  add_hx<wg_cfg<f, radix_list<16, 16, 16, 8>, 1024, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 1>>("hx 16.16.16.8 wg1024 TWL1");
  add_hx<wg_cfg<f, radix_list<16, 16, 16, 8>, 512, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>("hx 16.16.16.8 wg512 TWL1");
  add_hx<wg_cfg<f, radix_list<8, 16, 16, 16>, 1024, 1, 8, 1, TW_GLOBAL, 4, NT, 0, 2>>("hx 8.16.16.16 wg1024 TWL2");
  add_hx<wg_cfg<f, radix_list<32, 32, 32>, 1024, 1, 32, 1, TW_GLOBAL, 4, NT, 0, 0>>("hx 32.32.32 wg1024 TWL0");
#elif TUNE_CASE == 16384064  // fp64 16384: the same 256 KiB
  using T = d; const int N = 16384;
  add_hx<wg_cfg<d, radix_list<16, 32, 32>, 512, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.32.32 wg512 TWL1");
  add_hx<wg_cfg<d, radix_list<16, 16, 8, 8>, 512, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.16.8.8 wg512 TWL1");
  add_hx<wg_cfg<d, radix_list<16, 16, 8, 8>, 1024, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 1>>("f64 hx 16.16.8.8 wg1024 TWL1");
  add_hx<wg_cfg<d, radix_list<8, 8, 16, 16>, 512, 1, 8, 1, TW_GLOBAL, 2, NT, 0, 2>>("f64 hx 8.8.16.16 wg512 TWL2");
  add_hx<wg_cfg<d, radix_list<16, 32, 32>, 512, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.32.32 wg512 TWL1 (again)");
#elif TUNE_CASE == 8192064  // fp64 8192 (reference WorkgroupOrGlobal size): LDS-resident production entry against register-resident forms
  using T = d; const int N = 8192;
  add<wg_cfg_twl<d, radix_list<16, 8, 8, 8>, 512, 1, 16, 1, 2, NT>, false>("f64 16.8.8.8 wg512 (production)");
  add_hx<wg_cfg<d, radix_list<16, 16, 32>, 512, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.16.32 wg512");
  add_hx_pf<wg_cfg<d, radix_list<16, 16, 32>, 512, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.16.32 wg512 PF");
  add_hx_pf<wg_cfg<d, radix_list<32, 16, 16>, 512, 1, 32, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 32.16.16 wg512 PF");
  add_hx_pf<wg_cfg<d, radix_list<16, 32, 16>, 512, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.32.16 wg512 PF");
  add_hx_pf<wg_cfg<d, radix_list<16, 16, 32>, 256, 1, 16, 1, TW_GLOBAL, 1, NT, 0, 1>>("f64 hx 16.16.32 wg256 PF");
  add_hx<wg_cfg<d, radix_list<16, 16, 32>, 1024, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 1>>("f64 hx 16.16.32 wg1024");
#elif TUNE_CASE == 16387  // fp32 16384: the TW_REGS production entry against register-resident forms
  using T = f; const int N = 16384;
  add<wg_cfg<f, radix_list<32, 16, 32>, 512, 1, 0, 0, TW_REGS, 2, NT>, false>("r32.16.32 twR wg512 (production)");
  add_hx<wg_cfg<f, radix_list<32, 32, 16>, 1024, 1, 32, 1, TW_GLOBAL, 4, NT, 0, 1>>("hx 32.32.16 wg1024");
  add_hx_pf<wg_cfg<f, radix_list<32, 32, 16>, 1024, 1, 32, 1, TW_GLOBAL, 4, NT, 0, 1>>("hx 32.32.16 wg1024 PF");
  add_hx_pf<wg_cfg<f, radix_list<32, 32, 16>, 512, 1, 32, 1, TW_GLOBAL, 2, NT, 0, 1>>("hx 32.32.16 wg512 PF");
  add_hx_pf<wg_cfg<f, radix_list<32, 16, 32>, 512, 1, 32, 1, TW_GLOBAL, 2, NT, 0, 1>>("hx 32.16.32 wg512 PF");
  add_hx_pf<wg_cfg<f, radix_list<16, 32, 32>, 1024, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 1>>("hx 16.32.32 wg1024 PF");
#elif TUNE_CASE == 16388  // fp32 16384: register-resident forms with TWO work-groups per CU (half images of 64 KiB)
  using T = f; const int N = 16384;
  add<wg_cfg<f, radix_list<32, 16, 32>, 512, 1, 0, 0, TW_REGS, 2, NT>, false>("r32.16.32 twR wg512 (production)");
  add_hx<wg_cfg<f, radix_list<32, 32, 16>, 512, 1, 32, 1, TW_GLOBAL, 4, NT, 0, 1>>("hx 32.32.16 wg512 occ4");
  add_hx<wg_cfg<f, radix_list<32, 16, 32>, 512, 1, 32, 1, TW_GLOBAL, 4, NT, 0, 1>>("hx 32.16.32 wg512 occ4");
  add_hx<wg_cfg<f, radix_list<16, 32, 32>, 512, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 1>>("hx 16.32.32 wg512 occ4");
  add_hx<wg_cfg<f, radix_list<32, 32, 16>, 256, 1, 32, 1, TW_GLOBAL, 2, NT, 0, 1>>("hx 32.32.16 wg256 occ2");
  add_hx<wg_cfg<f, radix_list<16, 16, 8, 8>, 512, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 2>>("hx 16.16.8.8 wg512 occ4");
  add_hx<wg_cfg<f, radix_list<32, 32, 16>, 1024, 1, 32, 1, TW_GLOBAL, 8, NT, 0, 1>>("hx 32.32.16 wg1024 occ8");
#elif TUNE_CASE == 8192065  // fp64 8192: ... the same
  using T = d; const int N = 8192;
  add_hx_pf<wg_cfg<d, radix_list<16, 32, 16>, 512, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.32.16 wg512 PF (production)");
  add_hx<wg_cfg<d, radix_list<16, 32, 16>, 512, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 1>>("f64 hx 16.32.16 wg512 occ4");
  add_hx<wg_cfg<d, radix_list<16, 16, 32>, 512, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 1>>("f64 hx 16.16.32 wg512 occ4");
  add_hx<wg_cfg<d, radix_list<32, 16, 16>, 256, 1, 32, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 32.16.16 wg256 occ2");
  add_hx<wg_cfg<d, radix_list<16, 32, 16>, 256, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.32.16 wg256 occ2");
#elif TUNE_CASE == 8192066  // fp64 8192 pair: padding of the 128-bit exchange image (21 % conflict cycles with period 16)
  using T = d; const int N = 8192;
  add_hx<wg_cfg<d, radix_list<16, 32, 16>, 256, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.32.16 wg256 pad 16/1 (production)");
  add_hx<wg_cfg<d, radix_list<16, 32, 16>, 256, 1, 16, 2, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.32.16 wg256 pad 16/2");
  add_hx<wg_cfg<d, radix_list<16, 32, 16>, 256, 1, 32, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.32.16 wg256 pad 32/1");
  add_hx<wg_cfg<d, radix_list<16, 32, 16>, 256, 1, 8, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.32.16 wg256 pad 8/1");
  add_hx<wg_cfg<d, radix_list<16, 32, 16>, 256, 1, 0, 0, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 16.32.16 wg256 no pad");
  add_hx<wg_cfg<d, radix_list<32, 16, 16>, 256, 1, 32, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 32.16.16 wg256 pad 32/1");
  add_hx<wg_cfg<d, radix_list<32, 16, 16>, 256, 1, 32, 2, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 32.16.16 wg256 pad 32/2");
  add_hx<wg_cfg<d, radix_list<32, 16, 16>, 256, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>("f64 hx 32.16.16 wg256 pad 16/1");
#elif TUNE_CASE == 12000  // fp32 12000 (96 KB): LDS-resident planner choice against register-resident forms, two work-groups per CU
  using T = f; const int N = 12000;
  add<wg_cfg<f, radix_list<30, 20, 20>, 640, 1, 0, 0, TW_GLOBAL, 2, NT, 0, 1>, false>("r30.20.20 wg640 lds");
  add_hx<wg_cfg<f, radix_list<30, 20, 20>, 640, 1, 0, 0, TW_GLOBAL, 3, NT, 0, 1>>("hx 30.20.20 wg640 occ3");
  add_hx<wg_cfg<f, radix_list<30, 20, 20>, 448, 1, 0, 0, TW_GLOBAL, 4, NT, 0, 1>>("hx 30.20.20 wg448 occ4");
  add_hx<wg_cfg<f, radix_list<24, 25, 20>, 512, 1, 0, 0, TW_GLOBAL, 4, NT, 0, 1>>("hx 24.25.20 wg512 occ4");
  add_hx<wg_cfg<f, radix_list<20, 20, 30>, 640, 1, 0, 0, TW_GLOBAL, 3, NT, 0, 1>>("hx 20.20.30 wg640 occ3");
#elif TUNE_CASE == 16384
  using S = radix_list<32, 32, 16>; using T = f; const int N = 16384;
  add<wg_cfg<f, S, 512, 1, 16, 1, TW_GLOBAL, 2, NT>, false>("r32.32.16 twG wg512 o2");
  add<wg_cfg<f, radix_list<16, 16, 8, 8>, 1024, 1, 16, 1, TW_GLOBAL, 4, NT>, false>("r16.16.8.8 twG wg1024 o4");
  add<wg_cfg<f, radix_list<16, 16, 16, 4>, 1024, 1, 16, 1, TW_GLOBAL, 4, NT>, false>("r16.16.16.4 twG wg1024 o4");
  add<wg_cfg<f, radix_list<32, 32, 16>, 512, 1, 0, 0, TW_GLOBAL, 2, NT>, false>("r32.32.16 twG wg512 nopad");
#elif TUNE_CASE == 4096064
  using S = radix_list<16, 16, 16>; using T = d; const int N = 4096;
  add<wg_cfg<d, S, 256, 1, 16, 1, TW_GLOBAL, 2, NT>, false>("f64 twG o2");
  add<wg_cfg<d, S, 256, 1, 16, 1, TW_GLOBAL, 1, NT>, true>("f64 twG o1 PF");
  add<wg_cfg<d, S, 256, 1, 16, 1, TW_REGS, 1, NT>, false>("f64 twR o1");
  add<wg_cfg<d, S, 256, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>, false>("f64 twG TWL1 o2");
  add<wg_cfg<d, S, 256, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>, true>("f64 twG TWL1 o2 PF");
  add<wg_cfg<d, radix_list<8, 8, 8, 8>, 512, 1, 16, 1, TW_GLOBAL, 2, NT>, false>("f64 r8x4 wg512 o2");
#elif TUNE_CASE == 16385
  using S = radix_list<32, 32, 16>; using T = f; const int N = 16384;
  add<wg_cfg<f, S, 512, 1, 0, 0, TW_GLOBAL, 2, NT>, false>("r32.32.16 twG wg512 nopad");
  add<wg_cfg<f, S, 512, 1, 0, 0, TW_REGS, 2, NT>, false>("r32.32.16 twR wg512 nopad");
  add<wg_cfg<f, S, 512, 1, 16, 1, TW_REGS, 2, NT>, false>("r32.32.16 twR wg512 pad");
  add<wg_cfg<f, radix_list<32, 16, 32>, 512, 1, 0, 0, TW_REGS, 2, NT>, false>("r32.16.32 twR wg512 nopad");
#elif TUNE_CASE == 16386
  // prefetching (software-pipelined loads) forms of the one-work-group-per-CU lengths
  using S = radix_list<32, 16, 32>; using T = f; const int N = 16384;
  add<wg_cfg<f, S, 512, 1, 0, 0, TW_REGS, 2, NT>, false>("r32.16.32 twR wg512 (production)");
  add<wg_cfg<f, S, 512, 1, 0, 0, TW_REGS, 1, NT>, true>("r32.16.32 twR wg512 o1 PF");
  add<wg_cfg<f, S, 512, 1, 0, 0, TW_GLOBAL, 1, NT>, true>("r32.16.32 twG wg512 o1 PF");
  add<wg_cfg<f, S, 512, 1, 0, 0, TW_GLOBAL, 2, NT>, true>("r32.16.32 twG wg512 o2 PF");
  add<wg_cfg<f, radix_list<16, 32, 32>, 1024, 1, 0, 0, TW_GLOBAL, 1, NT>, true>("r16.32.32 twG wg1024 o1 PF");
  add<wg_cfg<f, radix_list<16, 32, 32>, 1024, 1, 0, 0, TW_GLOBAL, 1, NT>, false>("r16.32.32 twG wg1024 o1");
#elif TUNE_CASE == 8193
  using S = radix_list<32, 16, 16>; using T = f; const int N = 8192;
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 2, NT>, false>("r32.16.16 twR wg256 o2 (production)");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 2, NT>, true>("r32.16.16 twR wg256 o2 PF");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 2, NT>, true>("r32.16.16 twG wg256 o2 PF");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>, true>("r32.16.16 twG TWL1 o2 PF");
  add<wg_cfg<f, radix_list<16, 16, 32>, 512, 1, 16, 1, TW_GLOBAL, 2, NT>, true>("r16.16.32 twG wg512 o2 PF");
  add<wg_cfg<f, radix_list<16, 16, 32>, 512, 1, 16, 1, TW_GLOBAL, 2, NT>, false>("r16.16.32 twG wg512 o2");
#elif TUNE_CASE == 1024064
  using S = radix_list<16, 8, 8>; using T = d; const int N = 1024;
  add<wg_cfg<d, S, 256, 4, 16, 1, TW_GLOBAL, 2, NT>, false>("f64 1024 twG fpw4 o2");
  add<wg_cfg<d, S, 256, 4, 16, 1, TW_REGS, 2, NT>, false>("f64 1024 twR fpw4 o2");
  add<wg_cfg<d, S, 64, 1, 16, 1, TW_REGS, 2, NT>, false>("f64 1024 twR wg64 fpw1 o2");
  add<wg_cfg<d, S, 128, 2, 16, 1, TW_REGS, 2, NT>, false>("f64 1024 twR wg128 fpw2 o2");
#elif TUNE_CASE == 2048064
  using S = radix_list<16, 16, 8>; using T = d; const int N = 2048;
  add<wg_cfg<d, S, 256, 2, 16, 1, TW_GLOBAL, 2, NT>, false>("f64 2048 twG fpw2 o2");
  add<wg_cfg<d, S, 256, 2, 16, 1, TW_REGS, 2, NT>, false>("f64 2048 twR fpw2 o2");
  add<wg_cfg<d, S, 128, 1, 16, 1, TW_REGS, 2, NT>, false>("f64 2048 twR wg128 fpw1 o2");
#elif TUNE_CASE == 8192064
  using S = radix_list<16, 16, 16, 2>; using T = d; const int N = 8192;
  add<wg_cfg<d, S, 512, 1, 16, 1, TW_GLOBAL, 2, NT>, false>("f64 8192 r16.16.16.2 twG wg512");
  add<wg_cfg<d, radix_list<16, 8, 8, 8>, 512, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 2>, false>("f64 8192 r16.8.8.8 TWL2 wg512");
  add<wg_cfg<d, radix_list<16, 8, 8, 8>, 512, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>, false>("f64 8192 r16.8.8.8 TWL1 wg512");
  add<wg_cfg<d, radix_list<32, 16, 16>, 256, 1, 16, 1, TW_GLOBAL, 1, NT>, false>("f64 8192 r32.16.16 twG wg256 o1");
  add<wg_cfg<d, radix_list<32, 16, 16>, 256, 1, 0, 0, TW_GLOBAL, 1, NT>, false>("f64 8192 r32.16.16 twG wg256 nopad");
  add<wg_cfg<d, radix_list<16, 8, 8, 8>, 512, 1, 16, 1, TW_GLOBAL, 2, NT>, false>("f64 8192 r16.8.8.8 twG wg512");
#elif TUNE_CASE == 64
  using S = radix_list<8, 8>; using T = f; const int N = 64;
  add<wg_cfg_twl<f, S, 256, 32, 8, 1, 4, NT, 1>, false>("64 STAGED (production)");
  add_xlane<wg_cfg<f, S, 256, 32, 8, 1, TW_GLOBAL, 4, NT, 1>>("64 cross-lane wg256");
  add_xlane<wg_cfg<f, S, 256, 32, 0, 0, TW_GLOBAL, 4, NT, 1>>("64 cross-lane wg256 nopad");
  add_xlane<wg_cfg<f, S, 512, 64, 8, 1, TW_GLOBAL, 4, NT, 1>>("64 cross-lane wg512");
#elif TUNE_CASE == 257
  using S = radix_list<16, 16>; using T = f; const int N = 256;
  add<wg_cfg_twl<f, S, 256, 16, 16, 1, 4, NT>, false>("256 direct I/O (production)");
  add<wg_cfg_twl<f, S, 256, 16, 16, 1, 4, NT, 1>, false>("256 STAGED");
  add_xlane<wg_cfg<f, S, 256, 16, 16, 1, TW_GLOBAL, 4, NT, 1>>("256 cross-lane wg256");
  add_xlane<wg_cfg<f, S, 256, 16, 0, 0, TW_GLOBAL, 4, NT, 1>>("256 cross-lane wg256 nopad");
  add_xlane<wg_cfg<f, S, 256, 16, 16, 1, TW_GLOBAL, 2, NT, 1>>("256 cross-lane wg256 occ2");
#elif TUNE_CASE == 5121
  using S = radix_list<8, 8, 8>; using T = f; const int N = 512;
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 4, NT, 0, 2>, false>("512 direct TWL2 (production)");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 4, NT, 1, 2>, false>("512 STAGED TWL2");
  add<wg_cfg<f, S, 256, 8, 16, 1, TW_GLOBAL, 4, NT, 1, 2>, false>("512 STAGED TWL2 fpw8 (32 lanes per FFT)");
#elif TUNE_CASE == 10241
  using S = radix_list<16, 8, 8>; using T = f; const int N = 1024;
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 4, NT, 0, 2>, false>("1024 direct TWL2 (production)");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 4, NT, 1, 2>, false>("1024 STAGED TWL2");
#elif TUNE_CASE == 25664
  using S = radix_list<16, 16>; using T = d; const int N = 256;
  add<wg_cfg_twl<d, S, 256, 16, 16, 1, 2, NT>, false>("f64 256 direct (production)");
  add<wg_cfg_twl<d, S, 256, 16, 16, 1, 2, NT, 1>, false>("f64 256 STAGED");
  add<wg_cfg_twl<d, S, 128, 8, 16, 1, 2, NT, 1>, false>("f64 256 STAGED wg128 fpw8");
#elif TUNE_CASE == 192
  using S = radix_list<16, 12>; using T = f; const int N = 192;
  add<wg_cfg_twl<f, S, 256, 16, 16, 1, 4, NT>, false>("192 direct (production)");
  add<wg_cfg_twl<f, S, 256, 16, 16, 1, 4, NT, 1>, false>("192 STAGED");
#elif TUNE_CASE == 256
  using S = radix_list<16, 16>; using T = f; const int N = 256;
  add<wg_cfg<f, S, 256, 16, 16, 1, TW_GLOBAL, 4, NT>, false>("256 twG fpw16 o4");
  add<wg_cfg<f, S, 256, 16, 16, 1, TW_GLOBAL, 4, NT, 0, 1>, false>("256 twG fpw16 o4 TWL1");
  add<wg_cfg<f, S, 256, 16, 16, 1, TW_REGS, 4, NT>, false>("256 twR fpw16 o4");
  add<wg_cfg<f, S, 256, 16, 16, 1, TW_GLOBAL, 4, NT, 1>, false>("256 twG fpw16 o4 STAGED");
  add<wg_cfg<f, S, 64, 4, 16, 1, TW_REGS, 4, NT>, false>("256 twR wg64 fpw4 o4");
#elif TUNE_CASE == 4096
  using S = radix_list<16, 16, 16>; using T = f; const int N = 4096;
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 3, NT>, true>("twR o3 PF (current)");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 1>, false>("twG TWL1 o4");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 1>, true>("twG TWL1 o4 PF");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 3, NT, 0, 1>, true>("twG TWL1 o3 PF");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 4, NT>, false>("twR o4");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 4, NT>, true>("twG o4 PF");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 4, NT>, false>("twG o4");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 5, NT>, false>("twG o5");
#elif TUNE_CASE == 4098
  // cache-policy bits of the headline kernel's loads / stores (1 = sc0, 2 = nt, 16 = sc1; AUX = (stores + 1) << 8 | loads)
  using S = radix_list<16, 16, 16>; using T = f; const int N = 4096;
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 3, NT>, true>("loads nt, stores nt (current)");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 3, ((18 + 1) << 8) | 2>, true>("loads nt, stores nt|sc1");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 3, ((16 + 1) << 8) | 2>, true>("loads nt, stores sc1");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 3, ((19 + 1) << 8) | 2>, true>("loads nt, stores nt|sc0|sc1");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 3, ((2 + 1) << 8) | 18>, true>("loads nt|sc1, stores nt");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 3, ((2 + 1) << 8) | 3>, true>("loads nt|sc0, stores nt");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 3, ((18 + 1) << 8) | 18>, true>("loads nt|sc1, stores nt|sc1");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 3, ((3 + 1) << 8) | 2>, true>("loads nt, stores nt|sc0");
#elif TUNE_CASE == 1025
  using S = radix_list<16, 8, 8>; using T = f; const int N = 1024;
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 4, NT>, false>("1024 twG (current)");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 4, NT, 0, 1>, false>("1024 twG TWL1");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 4, NT, 0, 2>, false>("1024 twG TWL2");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_REGS, 4, NT>, false>("1024 twR");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_REGS, 3, NT>, true>("1024 twR o3 PF");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 3, NT, 0, 2>, true>("1024 twG TWL2 o3 PF");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 4, NT, 0, 2>, true>("1024 twG TWL2 o4 PF");
  add<wg_cfg<f, S, 64, 1, 16, 1, TW_REGS, 4, NT>, false>("1024 twR wg64 fpw1");
  add<wg_cfg<f, S, 128, 2, 16, 1, TW_REGS, 3, NT>, true>("1024 twR wg128 fpw2 o3 PF");
#elif TUNE_CASE == 3072
  using S = radix_list<16, 16, 12>; using T = f; const int N = 3072;
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 4, NT>, false>("3072 twG (current)");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 1>, false>("3072 twG TWL1");
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 2>, false>("3072 twG TWL2");
  add<wg_cfg<f, radix_list<16, 16, 12>, 192, 1, 16, 1, TW_GLOBAL, 4, NT, 0, 1>, false>("3072 wg192 TWL1");
#elif TUNE_CASE == 1000
  using S = radix_list<10, 10, 10>; using T = f; const int N = 1000;
  add<wg_cfg<f, S, 200, 2, 0, 0, TW_GLOBAL, 4, NT>, false>("1000 twG (current)");
  add<wg_cfg<f, S, 200, 2, 0, 0, TW_GLOBAL, 4, NT, 0, 1>, false>("1000 twG TWL1");
  add<wg_cfg<f, S, 200, 2, 0, 0, TW_GLOBAL, 4, NT, 0, 2>, false>("1000 twG TWL2");
#elif TUNE_CASE == 1024066
  using S = radix_list<16, 8, 8>; using T = d; const int N = 1024;
  add<wg_cfg<d, S, 256, 4, 16, 1, TW_GLOBAL, 2, NT>, false>("f64 1024 twG (current)");
  add<wg_cfg<d, S, 256, 4, 16, 1, TW_GLOBAL, 2, NT, 0, 1>, false>("f64 1024 TWL1");
  add<wg_cfg<d, S, 256, 4, 16, 1, TW_GLOBAL, 2, NT, 0, 2>, false>("f64 1024 TWL2");
#elif TUNE_CASE == 4097
  using S = radix_list<16, 16, 16>; using T = f; const int N = 4096;
  add<wg_cfg<f, S, 256, 1, 16, 1, TW_REGS, 3, NT>, true>("twR o3 PF (current)");
  add<wg_cfg<f, S, 512, 2, 16, 1, TW_REGS, 3, NT>, true>("twR wg512 fpw2 o3 PF");
  add<wg_cfg<f, S, 512, 2, 16, 1, TW_REGS, 4, NT>, false>("twR wg512 fpw2 o4");
  add<wg_cfg<f, S, 1024, 4, 16, 1, TW_REGS, 4, NT>, false>("twR wg1024 fpw4 o4");
  add<wg_cfg<f, S, 1024, 4, 16, 1, TW_REGS, 3, NT>, true>("twR wg1024 fpw4 o3 PF");
  add<wg_cfg<f, S, 128, 1, 16, 1, TW_REGS, 2, NT>, false>("twR wg128 (32pt) o2");
#elif TUNE_CASE == 512
  using S = radix_list<8, 8, 8>; using T = f; const int N = 512;
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 4, NT>, false>("twG fpw4 o4");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_REGS, 4, NT>, false>("twR fpw4 o4");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 4, NT, 0, 2>, false>("twG fpw4 o4 TWL2");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_REGS, 4, NT>, true>("twR fpw4 o4 PF");
  add<wg_cfg<f, radix_list<32, 16>, 256, 16, 16, 1, TW_GLOBAL, 2, NT>, false>("r32x16 twG fpw16 o2");
  add<wg_cfg<f, radix_list<16, 32>, 256, 16, 16, 1, TW_GLOBAL, 2, NT>, false>("r16x32 twG fpw16 o2");
  add<wg_cfg<f, S, 256, 4, 16, 1, TW_GLOBAL, 4, NT, 1>, false>("twG fpw4 o4 STAGED");
#endif
  const long long nfft = (long long)(bytes / (sizeof(cx<T>) * N));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct gridopt { const char* name; int mode; int k; };
#ifdef TUNE_GRID_SWEEP  // persistent grids of k / 4 x resident work-groups (one-work-group-per-CU kernels)
  const gridopt gopts[] = {{"1.0xres", 2, 4}, {"1.25x", 2, 5}, {"1.5x", 2, 6}, {"1.75x", 2, 7}, {"2xres", 2, 8}, {"2.5x", 2, 10}, {"3xres", 2, 12}, {"6xres", 2, 24}};
#else
  const gridopt gopts[] = {{"2xres", 0, 2}, {"4xres", 0, 4}, {"grp/8", 1, 8}, {"grp/4", 1, 4}, {"grp/2", 1, 2}, {"grp/1", 1, 1}};
#endif
  const int NGO = sizeof(gopts) / sizeof(gopts[0]);
  std::vector<std::vector<std::vector<float>>> times(g_variants.size(), std::vector<std::vector<float>>(NGO));
  for (int round = 0; round < 6; ++round) {
    for (size_t v = 0; v < g_variants.size(); ++v) {
      int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, g_variants[v].fn, g_variants[v].wg, g_variants[v].lds));
      const long long groups = (nfft + g_variants[v].fpw - 1) / g_variants[v].fpw;
      for (int go = 0; go < NGO; ++go) {
        long long grid = gopts[go].mode == 0 ? (long long)gopts[go].k * occ * cus
                         : gopts[go].mode == 2 ? (long long)gopts[go].k * occ * cus / 4 : (groups + gopts[go].k - 1) / gopts[go].k;
        grid = std::max<long long>(1, std::min(grid, groups));
        CK(hipEventRecord(e0));
        g_variants[v].launch((unsigned)grid, nfft);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (round) times[v][go].push_back(ms);
      }
    }
  }
  CK(hipGetLastError());
  for (size_t v = 0; v < g_variants.size(); ++v) {
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, g_variants[v].fn, g_variants[v].wg, g_variants[v].lds));
    printf("%-28s occ=%d lds=%-6zu", g_variants[v].name.c_str(), occ, g_variants[v].lds);
    for (int go = 0; go < NGO; ++go) {
      auto t = times[v][go]; std::sort(t.begin(), t.end());
      printf("  %s %.2f", gopts[go].name, 2.0 * bytes / t[t.size() / 2] * 1e-9);
    }
    printf("  TB/s\n");
  }
  {  // the variants agree with the first one (three sampled transforms, relative L2)
    const long long sample[3] = {0, nfft / 2 + 1, nfft - 1};
    std::vector<std::vector<cx<T>>> ref(3, std::vector<cx<T>>(N)), got(3, std::vector<cx<T>>(N));
    for (size_t v = 0; v < g_variants.size(); ++v) {
      int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, g_variants[v].fn, g_variants[v].wg, g_variants[v].lds));
      const long long groups = (nfft + g_variants[v].fpw - 1) / g_variants[v].fpw;
      CK(hipMemset(g_out, 0xff, bytes));
      g_variants[v].launch((unsigned)std::max<long long>(1, std::min<long long>(2ll * occ * cus, groups)), nfft);
      CK(hipDeviceSynchronize());
      double worst = 0;
      for (int k = 0; k < 3; ++k) {
        CK(hipMemcpy((v == 0 ? ref : got)[k].data(), (const cx<T>*)g_out + sample[k] * N, sizeof(cx<T>) * N, hipMemcpyDeviceToHost));
        if (v == 0) continue;
        double num = 0, den = 0;
        for (int i = 0; i < N; ++i) {
          const double dr = (double)got[k][i].re - (double)ref[k][i].re, di = (double)got[k][i].im - (double)ref[k][i].im;
          num += dr * dr + di * di; den += (double)ref[k][i].re * ref[k][i].re + (double)ref[k][i].im * ref[k][i].im;
        }
        const double e = std::sqrt(num / den);
        worst = (e == e && e > worst) || e != e ? (e != e ? 1e30 : e) : worst;
      }
      if (v != 0) printf("%-28s rel-L2 against the first variant %.2e%s\n", g_variants[v].name.c_str(), worst, worst > 1e-5 ? "   <-- DIFFERS" : "");
    }
  }
  return 0;
}
