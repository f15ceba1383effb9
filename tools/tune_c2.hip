// Headline-kernel experiments (fp32 N=4096, batch 65536): variants of stockham_wg_prefetch_kernel<16.16.16>
//   base   the production kernel
//   rot    the 16 pass-0 legs of a work-group are issued starting at leg 4 * (blockIdx & 3) (VERDICT r1 #5 iii: is the
//          32-KiB-row vs 4-KiB-granule copy gap channel aliasing at the 2-KiB leg stride?)
//   swz    XOR-swizzled LDS image (element e at e ^ ((e >> 4) & 15)) instead of the +1/16 padding (VERDICT r1 #5 i):
//          every scatter and gather of the three passes is bank-conflict free, 32 KiB instead of 34 KiB per FFT
//   swz4   the same with __launch_bounds__ for 4 work-groups per CU (no next-FFT register image: no prefetch)
// Outputs of every variant are compared bit for bit with the production kernel's.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/tune_c2.hip -o tools/tune_c2_bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "../portfft_amd/csrc/stockham_wg.hpp"
using namespace pfa;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

using Seq = radix_list<16, 16, 16>;
using Cfg = wg_cfg<float, Seq, 256, 1, 16, 1, TW_REGS, 3, 2>;
using T = float;
constexpr int N = 4096;

template <int ROT>
PFA_DEV void pass0_load_rot(const packed_io<T, N, 1, 2>& io, int tid, cx<T> (&v)[1][16]) {
  sfor<0, 16>([&](auto t_) PFA_LAMBDA {
    constexpr int t = (decltype(t_)::value + ROT) % 16;
    v[0][t] = io.load(io.in_off(0, tid), io.in_step(t * 256));
  });
}

PFA_DEV void load_twr(const cx<T>* __restrict__ tw, int tid, cx<T> (&twr)[Cfg::TWR_TOTAL]) {
  sfor<1, 3>([&](auto p_) PFA_LAMBDA {
    constexpr int p = decltype(p_)::value;
    constexpr int Ns = Seq::ns(p);
    const int q = tid % Ns;
    sfor<1, 16>([&](auto t_) PFA_LAMBDA {
      constexpr int t = decltype(t_)::value;
      twr[Cfg::twr_off(p) + (t - 1)] = tw[Seq::tw_off(p) + (t - 1) * Ns + q];
    });
  });
}

// ---- rot: production passes, rotated issue order of the pass-0 loads (whole body instantiated per rotation) ----
template <int ROT>
PFA_DEV void rot_body(const cx<T>* in, cx<T>* out, const cx<T>* __restrict__ tw, long long nfft, cx<T>* lds) {
  const int tid = threadIdx.x;
  cx<T> twr[Cfg::TWR_TOTAL];
  load_twr(tw, tid, twr);
  using IO = packed_io<T, N, 1, 2>;
  long long g = blockIdx.x;
  cx<T> cur[1][16], nxt[1][16];
  {
    const IO io0(in, out, g, nfft);
    pass0_load_rot<ROT>(io0, tid, cur);
  }
  for (; g < nfft; g += gridDim.x) {
    const IO io(in, out, g, nfft);
    wg_pass0_compute<Cfg>(cur, lds, tid);
    const long long gn = g + gridDim.x;
    if (gn < nfft) {
      const IO ion(in, out, gn, nfft);
      pass0_load_rot<ROT>(ion, tid, nxt);
    }
    wg_passes<Cfg, false, 1>(io, 0u, lds, tid, tw, twr, T(1));
    sfor<0, 16>([&](auto t_) PFA_LAMBDA { cur[0][decltype(t_)::value] = nxt[0][decltype(t_)::value]; });
  }
}
__global__ __launch_bounds__(256, 3) void rot_kernel(const cx<T>* in, cx<T>* out, const cx<T>* __restrict__ tw,
                                                     long long nfft) {
  extern __shared__ __attribute__((aligned(16))) char pfa_smem[];
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem);
  if (static_cast<long long>(blockIdx.x) >= nfft) return;
  const unsigned rot = blockIdx.x & 3u;
  if (rot == 0) rot_body<0>(in, out, tw, nfft, lds);
  else if (rot == 1) rot_body<4>(in, out, tw, nfft, lds);
  else if (rot == 2) rot_body<8>(in, out, tw, nfft, lds);
  else rot_body<12>(in, out, tw, nfft, lds);
}

// ---- stag: long-lived work-groups (persistent grid) with a staggered start instead of many short-lived ones ----
// production de-phases the work-groups by giving each only 4 FFTs (16384 work-groups); every work-group lifetime
// then pays its twiddle loads and one un-prefetched first load.  Here: grid = k x resident, work-group i sleeps
// (i % PHASES) * step before its first load.
__global__ __launch_bounds__(256, 3) void stag_kernel(const cx<T>* in, cx<T>* out, const cx<T>* __restrict__ tw,
                                                      long long nfft, int phases, int step) {
  extern __shared__ __attribute__((aligned(16))) char pfa_smem[];
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem);
  if (static_cast<long long>(blockIdx.x) >= nfft) return;
  const int ph = static_cast<int>(blockIdx.x % static_cast<unsigned>(phases));
  for (int i = 0; i < ph * step; ++i) __builtin_amdgcn_s_sleep(32);  // 32 x 64 cycles ~ 0.85 us at 2.4 GHz
  rot_body<0>(in, out, tw, nfft, lds);
}

// ---- tail: two-tier grid -- n_main work-groups with 4 FFTs each, then short work-groups (tail_k FFTs each) over the
// remaining FFTs, so that the end of the launch is balanced at a finer grain ----
template <int ROT>
PFA_DEV void range_body(const cx<T>* in, cx<T>* out, const cx<T>* __restrict__ tw, long long first, long long stride,
                        long long end, long long nfft, cx<T>* lds) {
  const int tid = threadIdx.x;
  cx<T> twr[Cfg::TWR_TOTAL];
  load_twr(tw, tid, twr);
  using IO = packed_io<T, N, 1, 2>;
  long long g = first;
  cx<T> cur[1][16], nxt[1][16];
  {
    const IO io0(in, out, g, nfft);
    pass0_load_rot<ROT>(io0, tid, cur);
  }
  for (; g < end; g += stride) {
    const IO io(in, out, g, nfft);
    wg_pass0_compute<Cfg>(cur, lds, tid);
    const long long gn = g + stride;
    if (gn < end) {
      const IO ion(in, out, gn, nfft);
      pass0_load_rot<ROT>(ion, tid, nxt);
    }
    wg_passes<Cfg, false, 1>(io, 0u, lds, tid, tw, twr, T(1));
    sfor<0, 16>([&](auto t_) PFA_LAMBDA { cur[0][decltype(t_)::value] = nxt[0][decltype(t_)::value]; });
  }
}
__global__ __launch_bounds__(256, 3) void tail_kernel(const cx<T>* in, cx<T>* out, const cx<T>* __restrict__ tw,
                                                      long long nfft, long long n_main, int tail_k) {
  extern __shared__ __attribute__((aligned(16))) char pfa_smem[];
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem);
  const long long b = blockIdx.x;
  if (b < n_main) {
    range_body<0>(in, out, tw, b, n_main, 4 * n_main, nfft, lds);
  } else {
    const long long n_tail = (nfft - 4 * n_main + tail_k - 1) / tail_k;  // tail work-groups
    const long long t = b - n_main;
    if (t < n_tail) range_body<0>(in, out, tw, 4 * n_main + t, n_tail, nfft, nfft, lds);
  }
}

// ---- swz: XOR-swizzled LDS image ----
template <bool PREFETCH, int OCC>
__global__ __launch_bounds__(256, OCC) void swz_kernel(const cx<T>* in, cx<T>* out, const cx<T>* __restrict__ tw,
                                                       long long nfft) {
  extern __shared__ __attribute__((aligned(16))) char pfa_smem[];
  char* lds = pfa_smem;  // 4096 elements x 8 B, element e at slot e ^ ((e >> 4) & 15)
  const unsigned tid = threadIdx.x;
  cx<T> twr[Cfg::TWR_TOTAL];
  load_twr(tw, static_cast<int>(tid), twr);
  using IO = packed_io<T, N, 1, 2>;
  // byte addresses (see the header comment of each pass)
  const unsigned s0 = ((16u * tid) | (tid & 15u)) * 8u;                       // pass-0 scatter: s0 ^ (u * 8)
  const unsigned g12 = (tid ^ ((tid >> 4) & 15u)) * 8u;                       // pass-1/2 gather: g12 + t * 2048
  const unsigned s1 = ((tid / 16u) * 256u + (tid & 15u)) * 8u;                // pass-1 scatter: (s1 ^ (u * 8)) + u * 128
  long long g = blockIdx.x;
  if (g >= nfft) return;
  cx<T> cur[1][16];
  [[maybe_unused]] cx<T> nxt[1][16];
  if constexpr (PREFETCH) {
    const IO io0(in, out, g, nfft);
    pass0_load_rot<0>(io0, static_cast<int>(tid), cur);
  }
  for (; g < nfft; g += gridDim.x) {
    const IO io(in, out, g, nfft);
    if constexpr (!PREFETCH) pass0_load_rot<0>(io, static_cast<int>(tid), cur);
    // pass 0: butterfly j = tid on inputs tid + 256 t; outputs e = 16 tid + u -> slot 16 tid + (u ^ (tid & 15))
    dft<16>(cur[0]);
    sfor<0, 16>([&](auto u_) PFA_LAMBDA {
      constexpr unsigned u = decltype(u_)::value;
      *reinterpret_cast<cx<T>*>(lds + (s0 ^ (u * 8u))) = cur[0][u];
    });
    __syncthreads();
    if constexpr (PREFETCH) {
      const long long gn = g + gridDim.x;
      if (gn < nfft) {
        const IO ion(in, out, gn, nfft);
        pass0_load_rot<0>(ion, static_cast<int>(tid), nxt);
      }
    }
    // pass 1 (Ns = 16): inputs e = tid + 256 t -> slot (tid ^ ((tid >> 4) & 15)) + 256 t
    cx<T> v[16];
    sfor<0, 16>([&](auto t_) PFA_LAMBDA {
      constexpr unsigned t = decltype(t_)::value;
      v[t] = *reinterpret_cast<const cx<T>*>(lds + g12 + t * 2048u);
    });
    __syncthreads();
    sfor<1, 16>([&](auto t_) PFA_LAMBDA {
      constexpr int t = decltype(t_)::value;
      v[t] = cmul(v[t], twr[Cfg::twr_off(1) + (t - 1)]);
    });
    dft<16>(v);
    // outputs e = (tid / 16) * 256 + tid % 16 + 16 u: (e >> 4) & 15 = u -> slot base + ((tid % 16) ^ u) + 16 u
    sfor<0, 16>([&](auto u_) PFA_LAMBDA {
      constexpr unsigned u = decltype(u_)::value;
      *reinterpret_cast<cx<T>*>(lds + (s1 ^ (u * 8u)) + u * 128u) = v[u];
    });
    __syncthreads();
    // pass 2 (Ns = 256): same gather; outputs tid + 256 u go to HBM in natural order
    sfor<0, 16>([&](auto t_) PFA_LAMBDA {
      constexpr unsigned t = decltype(t_)::value;
      v[t] = *reinterpret_cast<const cx<T>*>(lds + g12 + t * 2048u);
    });
    __syncthreads();
    sfor<1, 16>([&](auto t_) PFA_LAMBDA {
      constexpr int t = decltype(t_)::value;
      v[t] = cmul(v[t], twr[Cfg::twr_off(2) + (t - 1)]);
    });
    dft<16>(v);
    sfor<0, 16>([&](auto u_) PFA_LAMBDA {
      constexpr int u = decltype(u_)::value;
      io.store(v[u], io.out_off(0, tid), io.out_step(u * 256));
    });
    if constexpr (PREFETCH) {
      sfor<0, 16>([&](auto t_) PFA_LAMBDA { cur[0][decltype(t_)::value] = nxt[0][decltype(t_)::value]; });
    }
  }
}

// ---- dma: the next FFT is prefetched by LDS-DMA (buffer_load_dwordx4 ... lds) into a second, raw LDS image instead
// of a second register image (VERDICT r1 #5 ii); working image XOR-swizzled as in swz.  64 KiB per work-group.
template <int OCC>
__global__ __launch_bounds__(256, OCC) void dma_kernel(const cx<T>* in, cx<T>* out, const cx<T>* __restrict__ tw,
                                                       long long nfft) {
  extern __shared__ __attribute__((aligned(16))) char pfa_smem[];
  char* raw = pfa_smem;           // natural order, filled by DMA
  char* lds = pfa_smem + 32768;   // working image, element e at slot e ^ ((e >> 4) & 15)
  const unsigned tid = threadIdx.x;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid / 64u);
  const unsigned lane = tid % 64u;
  cx<T> twr[Cfg::TWR_TOTAL];
  load_twr(tw, static_cast<int>(tid), twr);
  using IO = packed_io<T, N, 1, 2>;
  const unsigned s0 = ((16u * tid) | (tid & 15u)) * 8u;
  const unsigned g12 = (tid ^ ((tid >> 4) & 15u)) * 8u;
  const unsigned s1 = ((tid / 16u) * 256u + (tid & 15u)) * 8u;
  auto issue_dma = [&](long long g) PFA_LAMBDA {
    const IO io(in, out, g, nfft);
    sfor<0, 8>([&](auto k_) PFA_LAMBDA {
      constexpr unsigned k = decltype(k_)::value;
      // each wave-instruction moves 1 KiB: lane l's 16 bytes land at the uniform LDS address + 16 l
      __builtin_amdgcn_raw_ptr_buffer_load_lds(io.rin, (__attribute__((address_space(3))) void*)(raw + wave * 8192u + k * 1024u),
                                               16, wave * 8192u + k * 1024u + lane * 16u, 0, 0, 2);
    });
  };
  long long g = blockIdx.x;
  if (g >= nfft) return;
  issue_dma(g);
  for (; g < nfft; g += gridDim.x) {
    const IO io(in, out, g, nfft);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave's DMA has landed; the working image is free (last reads were before the stores)
    cx<T> v[16];
    sfor<0, 16>([&](auto t_) PFA_LAMBDA {
      constexpr unsigned t = decltype(t_)::value;
      v[t] = *reinterpret_cast<const cx<T>*>(raw + tid * 8u + t * 2048u);
    });
    dft<16>(v);
    sfor<0, 16>([&](auto u_) PFA_LAMBDA {
      constexpr unsigned u = decltype(u_)::value;
      *reinterpret_cast<cx<T>*>(lds + (s0 ^ (u * 8u))) = v[u];
    });
    __syncthreads();  // raw image consumed by every wave
    const long long gn = g + gridDim.x;
    if (gn < nfft) issue_dma(gn);
    sfor<0, 16>([&](auto t_) PFA_LAMBDA {
      constexpr unsigned t = decltype(t_)::value;
      v[t] = *reinterpret_cast<const cx<T>*>(lds + g12 + t * 2048u);
    });
    __syncthreads();
    sfor<1, 16>([&](auto t_) PFA_LAMBDA {
      constexpr int t = decltype(t_)::value;
      v[t] = cmul(v[t], twr[Cfg::twr_off(1) + (t - 1)]);
    });
    dft<16>(v);
    sfor<0, 16>([&](auto u_) PFA_LAMBDA {
      constexpr unsigned u = decltype(u_)::value;
      *reinterpret_cast<cx<T>*>(lds + (s1 ^ (u * 8u)) + u * 128u) = v[u];
    });
    __syncthreads();
    sfor<0, 16>([&](auto t_) PFA_LAMBDA {
      constexpr unsigned t = decltype(t_)::value;
      v[t] = *reinterpret_cast<const cx<T>*>(lds + g12 + t * 2048u);
    });
    sfor<1, 16>([&](auto t_) PFA_LAMBDA {
      constexpr int t = decltype(t_)::value;
      v[t] = cmul(v[t], twr[Cfg::twr_off(2) + (t - 1)]);
    });
    dft<16>(v);
    sfor<0, 16>([&](auto u_) PFA_LAMBDA {
      constexpr int u = decltype(u_)::value;
      io.store(v[u], io.out_off(0, tid), io.out_step(u * 256));
    });
  }
}

struct variant { std::string name; size_t lds; const void* fn; std::function<void(unsigned)> launch; };

int main() {
  const long long nfft = 65536;
  const size_t bytes = (size_t)nfft * N * sizeof(cx<T>);
  void *d_in[2], *d_out, *d_ref;
  for (auto& p : d_in) CK(hipMalloc(&p, bytes));
  CK(hipMalloc(&d_out, bytes)); CK(hipMalloc(&d_ref, bytes));
  {
    std::vector<cx<T>> h((size_t)N * 1024);
    unsigned s = 1;
    for (auto& e : h) { s = s * 1664525u + 1013904223u; e.re = (T)((s >> 8) & 0xFFFF) / 32768 - 1; s = s * 1664525u + 1013904223u; e.im = (T)((s >> 8) & 0xFFFF) / 32768 - 1; }
    for (auto& p : d_in) for (size_t off = 0; off < bytes; off += h.size() * sizeof(cx<T>)) CK(hipMemcpy((char*)p + off, h.data(), h.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  }
  std::vector<cx<T>> tw(Seq::tw_total);
  for (int p = 1; p < 3; ++p) {
    const int Ns = Seq::ns(p);
    for (int t = 1; t < 16; ++t) for (int q = 0; q < Ns; ++q) {
      const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)(t * q) / (long double)(Ns * 16);
      tw[Seq::tw_off(p) + (t - 1) * Ns + q] = {(T)cosl(a), (T)sinl(a)};
    }
  }
  cx<T>* d_tw; CK(hipMalloc(&d_tw, tw.size() * sizeof(cx<T>))); CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  int which = 0;
  std::vector<variant> vs;
  auto in_ptr = [&]() { return (const cx<T>*)d_in[which & 1]; };
  vs.push_back({"base (production prefetch kernel)", Cfg::LDS_BYTES, (const void*)&stockham_wg_prefetch_kernel<Cfg, false>, [&](unsigned grid) {
    hipLaunchKernelGGL((stockham_wg_prefetch_kernel<Cfg, false>), dim3(grid), dim3(256), Cfg::LDS_BYTES, 0, in_ptr(), (cx<T>*)d_out, d_tw, nfft, 1.0f, 0ll, 4); }});
  vs.push_back({"rot (legs start at 4*(wg&3))", Cfg::LDS_BYTES, (const void*)&rot_kernel, [&](unsigned grid) {
    hipLaunchKernelGGL(rot_kernel, dim3(grid), dim3(256), Cfg::LDS_BYTES, 0, in_ptr(), (cx<T>*)d_out, d_tw, nfft); }});
  vs.push_back({"swz (XOR-swizzled LDS, prefetch, 3 WG/CU)", 32768, (const void*)&swz_kernel<true, 3>, [&](unsigned grid) {
    hipLaunchKernelGGL((swz_kernel<true, 3>), dim3(grid), dim3(256), 32768, 0, in_ptr(), (cx<T>*)d_out, d_tw, nfft); }});
  vs.push_back({"swz2 (XOR-swizzled LDS, prefetch, VGPR budget 256)", 32768, (const void*)&swz_kernel<true, 2>, [&](unsigned grid) {
    hipLaunchKernelGGL((swz_kernel<true, 2>), dim3(grid), dim3(256), 32768, 0, in_ptr(), (cx<T>*)d_out, d_tw, nfft); }});
  vs.push_back({"swz4 (XOR-swizzled LDS, no prefetch, 4 WG/CU)", 32768, (const void*)&swz_kernel<false, 4>, [&](unsigned grid) {
    hipLaunchKernelGGL((swz_kernel<false, 4>), dim3(grid), dim3(256), 32768, 0, in_ptr(), (cx<T>*)d_out, d_tw, nfft); }});
  CK(hipFuncSetAttribute((const void*)&dma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  vs.push_back({"dma (LDS-DMA prefetch into a raw image, 2 WG/CU)", 65536, (const void*)&dma_kernel<2>, [&](unsigned grid) {
    hipLaunchKernelGGL((dma_kernel<2>), dim3(grid), dim3(256), 65536, 0, in_ptr(), (cx<T>*)d_out, d_tw, nfft); }});
  {  // cache policies of the production kernel (was nt chosen on constant data in round 1?)
    using C0 = wg_cfg<float, Seq, 256, 1, 16, 1, TW_REGS, 3, 0>;
    using Cw = wg_cfg<float, Seq, 256, 1, 16, 1, TW_REGS, 3, 0x102>;   // loads nt, stores default
    using Cr = wg_cfg<float, Seq, 256, 1, 16, 1, TW_REGS, 3, 0x300>;   // loads default, stores nt
    vs.push_back({"policy: default loads + default stores", C0::LDS_BYTES, (const void*)&stockham_wg_prefetch_kernel<C0, false>, [&](unsigned grid) {
      hipLaunchKernelGGL((stockham_wg_prefetch_kernel<C0, false>), dim3(grid), dim3(256), C0::LDS_BYTES, 0, in_ptr(), (cx<T>*)d_out, d_tw, nfft, 1.0f, 0ll, 4); }});
    vs.push_back({"policy: nt loads + default stores", Cw::LDS_BYTES, (const void*)&stockham_wg_prefetch_kernel<Cw, false>, [&](unsigned grid) {
      hipLaunchKernelGGL((stockham_wg_prefetch_kernel<Cw, false>), dim3(grid), dim3(256), Cw::LDS_BYTES, 0, in_ptr(), (cx<T>*)d_out, d_tw, nfft, 1.0f, 0ll, 4); }});
    vs.push_back({"policy: default loads + nt stores", Cr::LDS_BYTES, (const void*)&stockham_wg_prefetch_kernel<Cr, false>, [&](unsigned grid) {
      hipLaunchKernelGGL((stockham_wg_prefetch_kernel<Cr, false>), dim3(grid), dim3(256), Cr::LDS_BYTES, 0, in_ptr(), (cx<T>*)d_out, d_tw, nfft, 1.0f, 0ll, 4); }});
  }
  for (long long n_main : {12288ll}) for (int tail_k : {2}) {
    const long long n_tail = (nfft - 4 * n_main + tail_k - 1) / tail_k;
    char nm[96]; snprintf(nm, sizeof(nm), "tail: %lld x 4 FFTs + %lld x %d", n_main, n_tail, tail_k);
    vs.push_back({nm, Cfg::LDS_BYTES, (const void*)&tail_kernel, [&, n_main, n_tail, tail_k](unsigned) {
      hipLaunchKernelGGL(tail_kernel, dim3((unsigned)(n_main + n_tail)), dim3(256), Cfg::LDS_BYTES, 0, in_ptr(), (cx<T>*)d_out, d_tw, nfft, n_main, tail_k); }});
  }
  for (int mult : {1}) for (int step : {0}) {
    const unsigned g = 768u * mult;
    char nm[96]; snprintf(nm, sizeof(nm), "stag grid %u (x%d resident) 16 phases step %d", g, mult, step);
    vs.push_back({nm, Cfg::LDS_BYTES, (const void*)&stag_kernel, [&, g, step](unsigned) {
      hipLaunchKernelGGL(stag_kernel, dim3(g), dim3(256), Cfg::LDS_BYTES, 0, in_ptr(), (cx<T>*)d_out, d_tw, nfft, 16, step); }});
  }
  // correctness: bit-identical to the production kernel
  vs[0].launch(16384); CK(hipDeviceSynchronize());
  CK(hipMemcpy(d_ref, d_out, bytes, hipMemcpyDeviceToDevice));
  std::vector<char> a((size_t)64 << 20), b((size_t)64 << 20);
  for (size_t v = 1; v < vs.size(); ++v) {
    CK(hipMemset(d_out, 0, bytes));
    vs[v].launch(16384); CK(hipDeviceSynchronize()); CK(hipGetLastError());
    bool same = true;
    for (size_t off : {(size_t)0, bytes / 2, bytes - a.size()}) {
      CK(hipMemcpy(a.data(), (char*)d_ref + off, a.size(), hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), (char*)d_out + off, b.size(), hipMemcpyDeviceToHost));
      same = same && std::memcmp(a.data(), b.data(), a.size()) == 0;
    }
    printf("check %-48s %s\n", vs[v].name.c_str(), same ? "bit-identical to base" : "DIFFERS");
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grids[] = {8192, 16384, 32768};
  std::vector<std::vector<std::vector<float>>> times(vs.size(), std::vector<std::vector<float>>(3));
  for (int round = 0; round < 12; ++round)
    for (size_t v = 0; v < vs.size(); ++v)
      for (int gi = 0; gi < 3; ++gi) {
        ++which;
        CK(hipEventRecord(e0)); vs[v].launch(grids[gi]); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (round >= 2) times[v][gi].push_back(ms);
      }
  CK(hipGetLastError());
  for (size_t v = 0; v < vs.size(); ++v) {
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vs[v].fn, 256, vs[v].lds));
    printf("%-48s occ=%d", vs[v].name.c_str(), occ);
    for (int gi = 0; gi < 3; ++gi) { auto t = times[v][gi]; std::sort(t.begin(), t.end()); printf("  grid %u: %.1f us %.2f TB/s (min %.1f)", grids[gi], t[t.size() / 2] * 1e3, 2.0 * bytes / t[t.size() / 2] * 1e-9, t[0] * 1e3); }
    printf("\n");
  }
  return 0;
}
