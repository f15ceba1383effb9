// Scratch tuner for the strided kernel on the C3 stage-A shape (fp64, 128 matrices of 1024 x 1024, column FFTs)
// and the C5 column pass (fp32, 256 matrices 1024 x 1024).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include "probes/stockham_strided_hx.hpp"
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
struct variant { std::string name; int fpw; int wg; size_t lds; const void* fn; std::function<void(unsigned)> launch; };
static std::vector<variant> g_variants;
static void *g_in, *g_out;
static long long g_total, g_inner, g_dist_outer; static unsigned g_stride; static int g_tiled = 0;

static void* g_stw_tab;  // store-modifier tables of M = 2^20: 3 levels x 128 entries (strided_args::stw_tab)
template <typename Cfg, int KIND, bool STW = false>
void add(const char* name) {
  using T = typename Cfg::T;
  auto tw = make_twiddles<typename Cfg::Seq, T>();
  cx<T>* d_tw; CK(hipMalloc(&d_tw, tw.size() * sizeof(cx<T>)));
  CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  const void* fn;
  if constexpr (KIND == 1) fn = (const void*)&stockham_strided_prefetch_kernel<Cfg, false, STW>;
  else if constexpr (KIND == 2) fn = (const void*)&stockham_strided_hx_kernel<Cfg, false, STW>;
  else fn = (const void*)&stockham_strided_kernel<Cfg, false, STW>;
  constexpr size_t lds = (KIND == 2 ? strided_hx_lds_bytes<Cfg>() : strided_lds_bytes<Cfg>()) + (STW ? 3 * 128 * sizeof(cx<T>) : 0);
  CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  g_variants.push_back({std::string(name) + (STW ? " +stw" : ""), Cfg::FPW, Cfg::WG, lds, fn, [d_tw](unsigned grid) {
    strided_args a{};
    a.in = g_in; a.out = g_out; a.tw = d_tw; a.total = g_total; a.inner = g_inner;
    a.in_dist_outer = a.out_dist_outer = g_dist_outer; a.in_stride = a.out_stride = g_stride; a.in_fdist = a.out_fdist = 1; a.scale = 1.0;
    a.stw_tab = g_stw_tab; a.stw_levels = 3; a.stw_lshift = 7; a.stw_cdiv = 1;
    if (g_tiled & 1) { a.in_gdist = (long long)Cfg::N * Cfg::FPW; a.in_stride = Cfg::FPW; }
    if (g_tiled & 2) { a.out_gdist = (long long)Cfg::N * Cfg::FPW; a.out_stride = Cfg::FPW; }
    if constexpr (KIND == 1) hipLaunchKernelGGL((stockham_strided_prefetch_kernel<Cfg, false, STW>), dim3(grid), dim3(Cfg::WG), lds, 0, a);
    else if constexpr (KIND == 2) hipLaunchKernelGGL((stockham_strided_hx_kernel<Cfg, false, STW>), dim3(grid), dim3(Cfg::WG), lds, 0, a);
    else hipLaunchKernelGGL((stockham_strided_kernel<Cfg, false, STW>), dim3(grid), dim3(Cfg::WG), lds, 0, a);
  }});
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  CK(hipMalloc(&g_in, bytes)); CK(hipMalloc(&g_out, bytes)); {  // random input (a constant fill makes on-die paths look faster than they are: profiles/r2_notes.md)
    std::vector<unsigned> h((size_t)1 << 22);
    unsigned sd = 12345u;
    for (auto& w : h) { sd = sd * 1664525u + 1013904223u; w = 0x3c000000u | (sd >> 9); }  // floats in [0.0078, 0.031)
    for (size_t off = 0; off < bytes; off += h.size() * 4) CK(hipMemcpy((char*)g_in + off, h.data(), std::min(h.size() * 4, bytes - off), hipMemcpyHostToDevice));
  }
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); const int cus = prop.multiProcessorCount;
  using f = float; using d = double; constexpr int NT = 2;
  {  // store-modifier tables of M = 2^20 (three levels of 128 entries), in the precision of the case
#if TUNE_CASE == 3 || TUNE_CASE == 4
    using TT = double;
#else
    using TT = float;
#endif
    std::vector<cx<TT>> tab(3 * 128);
    for (int l = 0; l < 3; ++l) for (int i = 0; i < 128; ++i) {
      const long double a = -2.0L * 3.14159265358979323846264338327950288L * (((long long)i << (7 * l)) % 1048576) / 1048576.0L;
      tab[l * 128 + i] = {(TT)cosl(a), (TT)sinl(a)};
    }
    CK(hipMalloc(&g_stw_tab, tab.size() * sizeof(tab[0]))); CK(hipMemcpy(g_stw_tab, tab.data(), tab.size() * sizeof(tab[0]), hipMemcpyHostToDevice));
  }
#if TUNE_CASE == 4
  // cache-policy bits of the loads of the C3 stage kernel (1 = sc0, 2 = nt, 16 = sc1; AUX = (stores + 1) << 8 | loads)
  using T = d; g_stride = 1024; g_inner = 1024; g_dist_outer = 1 << 20; g_total = 128 * 1024;
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 2, NT, 0, 1>, 0, true>("loads nt stores nt");
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 2, ((2 + 1) << 8) | 0, 0, 1>, 0, true>("loads plain stores nt");
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 2, ((2 + 1) << 8) | 1, 0, 1>, 0, true>("loads sc0 stores nt");
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 2, ((2 + 1) << 8) | 16, 0, 1>, 0, true>("loads sc1 stores nt");
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 2, ((2 + 1) << 8) | 18, 0, 1>, 0, true>("loads nt|sc1 stores nt");
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 2, ((2 + 1) << 8) | 3, 0, 1>, 0, true>("loads nt|sc0 stores nt");
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 2, ((18 + 1) << 8) | 2, 0, 1>, 0, true>("loads nt stores nt|sc1");
#elif TUNE_CASE == 5
  // fp32 N = 2^20 four-step stages (n = 1024 columns, 128-byte segments): stage A = strided in / tiled out + stw,
  // stage B = tiled in / strided out.  Production: 16.8.8 on 1024 lanes (stw stages), 32.32 prefetch (column/column)
  using T = f; g_stride = 1024; g_inner = 1024; g_dist_outer = 1 << 20; g_total = 256 * 1024;
  add<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, NT>, 0>("f32 16.8.8 wg1024 fpw16 (production stw)");
  add<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, NT>, 0, true>("f32 16.8.8 wg1024 fpw16 (production stw)");
  add<strided_cfg<f, radix_list<32, 32>, 512, 16, 2, NT>, 1>("PF f32 32.32 wg512 fpw16 (production c/c)");
  add<strided_cfg<f, radix_list<32, 32>, 512, 16, 2, NT>, 1, true>("PF f32 32.32 wg512 fpw16");
  add<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, NT>, 1>("PF f32 16.8.8 wg1024 fpw16");
  add<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, NT>, 1, true>("PF f32 16.8.8 wg1024 fpw16");
  add<strided_cfg<f, radix_list<16, 8, 8>, 512, 16, 4, NT>, 2>("HX f32 16.8.8 wg512(32pt) fpw16 2/CU");
  add<strided_cfg<f, radix_list<16, 8, 8>, 512, 16, 4, NT>, 2, true>("HX f32 16.8.8 wg512(32pt) fpw16 2/CU");
  add<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 8, NT>, 2>("HX f32 16.8.8 wg1024(16pt) fpw16 2/CU");
  add<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 8, NT>, 2, true>("HX f32 16.8.8 wg1024(16pt) fpw16 2/CU");
  add<strided_cfg<f, radix_list<32, 32>, 512, 16, 4, NT>, 2>("HX f32 32.32 wg512(32pt) fpw16 2/CU");
  add<strided_cfg<f, radix_list<32, 32>, 512, 16, 4, NT>, 2, true>("HX f32 32.32 wg512(32pt) fpw16 2/CU");
  add<strided_cfg<f, radix_list<16, 8, 8>, 1024, 32, 4, NT>, 2>("HX f32 16.8.8 wg1024(32pt) fpw32 1/CU");
  add<strided_cfg<f, radix_list<16, 8, 8>, 1024, 32, 4, NT>, 2, true>("HX f32 16.8.8 wg1024(32pt) fpw32 1/CU");
  add<strided_cfg<f, radix_list<32, 32>, 1024, 32, 4, NT>, 2>("HX f32 32.32 wg1024(32pt) fpw32 1/CU");
  add<strided_cfg<f, radix_list<32, 32>, 1024, 32, 4, NT>, 2, true>("HX f32 32.32 wg1024(32pt) fpw32 1/CU");
  add<strided_cfg<f, radix_list<8, 16, 8>, 512, 16, 4, NT>, 2, true>("HX f32 8.16.8 wg512(32pt) fpw16 2/CU");
#elif TUNE_CASE == 3
  using T = d; g_stride = 1024; g_inner = 1024; g_dist_outer = 1 << 20; g_total = 128 * 1024;
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 2, NT, 0, 1>, 0>("f64 16.8.8 wg512 fpw8 TWL1 (production)");
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 2, NT, 0, 1>, 0, true>("f64 16.8.8 wg512 fpw8 TWL1 (production)");
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 2, NT, 0, 1>, 2>("HX f64 16.8.8 wg512 fpw8 TWL1");
  add<wg_cfg<d, radix_list<16, 8, 8>, 256, 8, 0, 0, TW_GLOBAL, 1, NT, 0, 1>, 2>("HX f64 16.8.8 wg256(32pt) fpw8 TWL1");
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 16, 0, 0, TW_GLOBAL, 2, NT, 0, 1>, 2>("HX f64 16.8.8 wg512(32pt) fpw16 TWL1");
  // software-pipelined loads (next group's loads in flight during the passes)
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 1, NT, 0, 1>, 1>("PF f64 16.8.8 wg512 fpw8 TWL1");
  add<wg_cfg<d, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 1, NT, 0, 1>, 1, true>("PF f64 16.8.8 wg512 fpw8 TWL1");
  // one exchange instead of two
  add<wg_cfg<d, radix_list<32, 32>, 256, 8, 0, 0, TW_GLOBAL, 1, NT, 0, 1>, 0>("f64 32.32 wg256 fpw8 TWL1");
  add<wg_cfg<d, radix_list<32, 32>, 256, 8, 0, 0, TW_GLOBAL, 1, NT, 0, 1>, 0, true>("f64 32.32 wg256 fpw8 TWL1");
  add<wg_cfg<d, radix_list<32, 32>, 512, 8, 0, 0, TW_GLOBAL, 1, NT, 0, 1>, 0>("f64 32.32 wg512(16pt, half the lanes idle per pass?) fpw8");
  add<wg_cfg<d, radix_list<16, 8, 8>, 1024, 8, 0, 0, TW_GLOBAL, 2, NT, 0, 1>, 0>("f64 16.8.8 wg1024(8pt) fpw8 TWL1");
  add<wg_cfg<d, radix_list<16, 8, 8>, 1024, 8, 0, 0, TW_GLOBAL, 2, NT, 0, 1>, 0, true>("f64 16.8.8 wg1024(8pt) fpw8 TWL1");
#else
  using T = f; g_stride = 1024; g_inner = 1024; g_dist_outer = 1 << 20; g_total = 256 * 1024;
  add<wg_cfg<f, radix_list<16, 8, 8>, 1024, 16, 0, 0, TW_GLOBAL, 4, NT>, 0>("f32 16.8.8 wg1024 fpw16");
  add<wg_cfg<f, radix_list<16, 8, 8>, 1024, 16, 0, 0, TW_GLOBAL, 2, NT>, 1>("f32 16.8.8 wg1024 fpw16 PF");
  add<wg_cfg<f, radix_list<32, 32>, 512, 16, 0, 0, TW_GLOBAL, 2, NT>, 0>("f32 32.32 wg512 fpw16");
  add<wg_cfg<f, radix_list<32, 32>, 512, 16, 0, 0, TW_GLOBAL, 2, NT>, 1>("f32 32.32 wg512 fpw16 PF");
  add<wg_cfg<f, radix_list<32, 32>, 512, 16, 0, 0, TW_GLOBAL, 2, NT, 0, 1>, 1>("f32 32.32 wg512 fpw16 PF TWL1");
  add<wg_cfg<f, radix_list<16, 8, 8>, 1024, 16, 0, 0, TW_GLOBAL, 4, NT, 0, 2>, 0>("f32 16.8.8 wg1024 fpw16 TWL2");
  add<wg_cfg<f, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 4, NT>, 0>("f32 16.8.8 wg512 fpw8");
  add<wg_cfg<f, radix_list<16, 8, 8>, 512, 8, 0, 0, TW_GLOBAL, 2, NT>, 1>("f32 16.8.8 wg512 fpw8 PF");
  // half-exchange forms (half-size LDS image: two work-groups per CU at 16 columns)
  add<wg_cfg<f, radix_list<32, 32>, 512, 16, 0, 0, TW_GLOBAL, 2, NT>, 2>("HX f32 32.32 wg512 fpw16");
  add<wg_cfg<f, radix_list<16, 8, 8>, 1024, 16, 0, 0, TW_GLOBAL, 2, NT>, 2>("HX f32 16.8.8 wg1024 fpw16");
  add<wg_cfg<f, radix_list<16, 8, 8>, 512, 16, 0, 0, TW_GLOBAL, 2, NT>, 2>("HX f32 16.8.8 wg512(32pt) fpw16");
  add<wg_cfg<f, radix_list<32, 32>, 1024, 32, 0, 0, TW_GLOBAL, 1, NT>, 2>("HX f32 32.32 wg1024 fpw32");
  add<wg_cfg<f, radix_list<16, 8, 8>, 1024, 32, 0, 0, TW_GLOBAL, 1, NT>, 2>("HX f32 16.8.8 wg1024(32pt) fpw32");
  add<wg_cfg<f, radix_list<32, 32>, 512, 16, 0, 0, TW_GLOBAL, 2, NT>, 1, true>("f32 32.32 wg512 fpw16 PF");
#endif
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct gridopt { const char* name; int mode; int k; };
  const gridopt gopts[] = {{"1xres", 0, 1}, {"2xres", 0, 2}, {"4xres", 0, 4}, {"grp/4", 1, 4}, {"grp/2", 1, 2}, {"grp/1", 1, 1}};
  const int NGO = 6;
  std::vector<std::vector<std::vector<float>>> times(g_variants.size(), std::vector<std::vector<float>>(NGO));
  for (g_tiled = 0; g_tiled < 3; ++g_tiled) {
  for (auto& t : times) for (auto& u : t) u.clear();
  printf("---- tiled input %d, tiled output %d\n", g_tiled & 1, (g_tiled >> 1) & 1);
  for (int round = 0; round < 4; ++round)
    for (size_t v = 0; v < g_variants.size(); ++v) {
      int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, g_variants[v].fn, g_variants[v].wg, g_variants[v].lds));
      const long long groups = g_total / g_variants[v].fpw;
      for (int go = 0; go < NGO; ++go) {
        long long grid = gopts[go].mode == 0 ? (long long)gopts[go].k * std::max(occ, 1) * cus : (groups + gopts[go].k - 1) / gopts[go].k;
        grid = std::max<long long>(1, std::min(grid, groups));
        CK(hipEventRecord(e0)); g_variants[v].launch((unsigned)grid); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (round) times[v][go].push_back(ms);
      }
    }
  CK(hipGetLastError());
  for (size_t v = 0; v < g_variants.size(); ++v) {
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, g_variants[v].fn, g_variants[v].wg, g_variants[v].lds));
    printf("%-30s occ=%d lds=%-6zu", g_variants[v].name.c_str(), occ, g_variants[v].lds);
    for (int go = 0; go < NGO; ++go) { auto t = times[v][go]; std::sort(t.begin(), t.end()); printf("  %s %.2f", gopts[go].name, 2.0 * bytes / t[t.size() / 2] * 1e-9); }
    printf("  TB/s\n");
  }
  }
  return 0;
}
