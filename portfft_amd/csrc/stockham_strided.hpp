// Work-group tier for strided / batch-interleaved data: FPW FFTs side by side, lanes walk the FFT index first.
//
// Role in the reference: the BATCH_INTERLEAVED branches of the work-group / sub-group dispatchers
// (/root/reference/src/portfft/dispatcher/workgroup_dispatcher.hpp:148-229, subgroup_dispatcher.hpp:174-417), the
// per-dimension launches of N-D transforms (committed_descriptor_impl.hpp:932-948) and the strided sub-FFTs +
// store-modifier multiply of the GLOBAL tier (common/global.hpp:135-170).  Design (ours):
//   * a work-group owns FPW FFTs whose elements with equal index are adjacent in memory ("columns") or whose rows
//     are adjacent ("rows"); lane l works on FFT f = l % FPW and butterfly slot j = l / FPW, so one wave-instruction
//     touches FPW adjacent columns x (64 / FPW) consecutive elements: full 128-byte lines for FPW*elem = 128 B
//     (columns) or (64/FPW)*elem = 128 B (rows);
//   * the LDS image is [element][f] (f fastest): every LDS access of every pass is lane-contiguous, no padding needed;
//   * same Stockham pass structure, twiddle tables and butterflies as stockham_wg.hpp;
//   * optional store modifier: output element k of the FFT with inner index c is multiplied by W_M^{k*c} taken from
//     two small tables (hi/lo split) -- the inter-factor twiddles of the four-step decomposition.
#pragma once
#include "stockham_wg.hpp"
#include "strided_args.hpp"

namespace pfa {

template <typename Cfg, bool BWD, bool STW, int P>
PFA_DEV void strided_pass(__amdgpu_buffer_rsrc_t rin, __amdgpu_buffer_rsrc_t rout, const strided_args& a, unsigned f,
                          unsigned tid, bool live, long long c0, cx<typename Cfg::T>* lds,
                          const cx<typename Cfg::T>* __restrict__ tw) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  constexpr int R = Seq::r[P];
  constexpr int N = Cfg::N;
  constexpr int NB = N / R;
  constexpr int Ns = Seq::ns(P);
  constexpr int BPT = Cfg::bpt(P);
  constexpr bool ragged = (NB % Cfg::TPF) != 0;
  constexpr bool first = P == 0;
  constexpr bool last = P == Cfg::NP - 1;
  constexpr int FPW = Cfg::FPW;
  constexpr unsigned ES = sizeof(cx<T>);

  cx<T> v[BPT][R];
  sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    if (!ragged || j < NB) {
      if constexpr (first) {
        // dead lanes (FFT beyond the end) get an out-of-range offset: the buffer range check returns zeros
        const unsigned voff = live ? (f * a.in_fdist + j * a.in_stride) * ES : 0xFFFFFFF0u;
        sfor<0, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          cx<T> x = buf_load<T, Cfg::AUX>(rin, voff, static_cast<unsigned>(t * NB) * a.in_stride * ES);
          if constexpr (BWD) x.im = -x.im;
          v[i][t] = x;
        });
      } else {
        const cx<T>* p = lds + j * FPW + f;
        sfor<0, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          v[i][t] = p[t * NB * FPW];
        });
      }
    }
  });
  if constexpr (!first) __syncthreads();
  sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    if (!ragged || j < NB) {
      const unsigned q = j % Ns;
      if constexpr (!first) {
        sfor<1, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          const cx<T> w = (tw + Seq::tw_off(P) + (t - 1) * Ns)[q];
          v[i][t] = cmul(v[i][t], w);
        });
      }
      dft<R>(v[i]);
      const unsigned base = (j / Ns) * (Ns * R) + q;
      if constexpr (last) {
        const unsigned voff = live ? (f * a.out_fdist + base * a.out_stride) * ES : 0xFFFFFFF0u;
        const T scale = static_cast<T>(a.scale);
        sfor<0, R>([&](auto u_) PFA_LAMBDA {
          constexpr int u = decltype(u_)::value;
          cx<T> y = v[i][u];
          if constexpr (STW) {
            const unsigned long long m = static_cast<unsigned long long>(base + u * Ns) * static_cast<unsigned long long>(c0 + f);
            const cx<T> wl = static_cast<const cx<T>*>(a.stw_lo)[m & ((1ull << a.stw_shift) - 1)];
            const cx<T> wh = static_cast<const cx<T>*>(a.stw_hi)[m >> a.stw_shift];
            y = cmul(y, cmul(wl, wh));
          }
          if constexpr (BWD) y.im = -y.im;
          y.re *= scale;
          y.im *= scale;
          buf_store<T, Cfg::AUX>(y, rout, voff, static_cast<unsigned>(u * Ns) * a.out_stride * ES);
        });
      } else {
        cx<T>* p = lds + base * FPW + f;
        sfor<0, R>([&](auto u_) PFA_LAMBDA {
          constexpr int u = decltype(u_)::value;
          p[u * Ns * FPW] = v[i][u];
        });
      }
    }
  });
  if constexpr (!last) __syncthreads();
}

template <typename Cfg, bool BWD, bool STW, int P>
PFA_DEV void strided_passes(__amdgpu_buffer_rsrc_t rin, __amdgpu_buffer_rsrc_t rout, const strided_args& a,
                            unsigned f, unsigned tid, bool live, long long c0, cx<typename Cfg::T>* lds,
                            const cx<typename Cfg::T>* __restrict__ tw) {
  if constexpr (P < Cfg::NP) {
    strided_pass<Cfg, BWD, STW, P>(rin, rout, a, f, tid, live, c0, lds, tw);
    strided_passes<Cfg, BWD, STW, P + 1>(rin, rout, a, f, tid, live, c0, lds, tw);
  }
}

/// Pass 0 of the strided kernel split in two (loads / butterfly + scatter) for the prefetching variant.
template <typename Cfg, bool BWD>
PFA_DEV void strided_pass0_load(__amdgpu_buffer_rsrc_t rin, const strided_args& a, unsigned f, unsigned tid, bool live,
                                cx<typename Cfg::T> (&v)[Cfg::bpt(0)][Cfg::Seq::r[0]]) {
  using T = typename Cfg::T;
  constexpr int R = Cfg::Seq::r[0];
  constexpr int NB = Cfg::N / R;
  constexpr bool ragged = (NB % Cfg::TPF) != 0;
  constexpr unsigned ES = sizeof(cx<T>);
  sfor<0, Cfg::bpt(0)>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    if (!ragged || j < NB) {
      const unsigned voff = live ? (f * a.in_fdist + j * a.in_stride) * ES : 0xFFFFFFF0u;
      sfor<0, R>([&](auto t_) PFA_LAMBDA {
        constexpr int t = decltype(t_)::value;
        cx<T> x = buf_load<T, Cfg::AUX>(rin, voff, static_cast<unsigned>(t * NB) * a.in_stride * ES);
        if constexpr (BWD) x.im = -x.im;
        v[i][t] = x;
      });
    }
  });
}

template <typename Cfg>
PFA_DEV void strided_pass0_compute(cx<typename Cfg::T> (&v)[Cfg::bpt(0)][Cfg::Seq::r[0]], unsigned f, unsigned tid,
                                   cx<typename Cfg::T>* lds) {
  constexpr int R = Cfg::Seq::r[0];
  constexpr int NB = Cfg::N / R;
  constexpr bool ragged = (NB % Cfg::TPF) != 0;
  sfor<0, Cfg::bpt(0)>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    if (!ragged || j < NB) {
      dft<R>(v[i]);
      cx<typename Cfg::T>* p = lds + (j * R) * Cfg::FPW + f;
      sfor<0, R>([&](auto u_) PFA_LAMBDA {
        constexpr int u = decltype(u_)::value;
        p[u * Cfg::FPW] = v[i][u];
      });
    }
  });
  __syncthreads();
}

template <typename Cfg>
PFA_DEV void strided_group(const strided_args& a, long long g, unsigned f, __amdgpu_buffer_rsrc_t* rin,
                           __amdgpu_buffer_rsrc_t* rout, bool* live, long long* c0_out) {
  using T = typename Cfg::T;
  constexpr unsigned ES = sizeof(cx<T>);
  const long long t0 = g * Cfg::FPW;
  const long long o = t0 / a.inner;
  const long long c0 = t0 - o * a.inner;
  *live = static_cast<long long>(f) < a.total - t0;
  *c0_out = c0;
  const cx<T>* in0 = static_cast<const cx<T>*>(a.in) + o * a.in_dist_outer + c0 * a.in_fdist;
  cx<T>* out0 = static_cast<cx<T>*>(a.out) + o * a.out_dist_outer + c0 * a.out_fdist;
  const unsigned in_bytes =
      (static_cast<unsigned>(Cfg::FPW - 1) * a.in_fdist + static_cast<unsigned>(Cfg::N - 1) * a.in_stride + 1) * ES;
  const unsigned out_bytes =
      (static_cast<unsigned>(Cfg::FPW - 1) * a.out_fdist + static_cast<unsigned>(Cfg::N - 1) * a.out_stride + 1) * ES;
  *rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<cx<T>*>(in0), 0, in_bytes, 0x00020000);
  *rout = __builtin_amdgcn_make_buffer_rsrc(out0, 0, out_bytes, 0x00020000);
}

/// Software-pipelined strided kernel: the loads of the work-group's next group are in flight during the LDS passes
/// of the current one (see stockham_wg_prefetch_kernel).
template <typename Cfg, bool BWD, bool STW>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_strided_prefetch_kernel(const strided_args a) {
  using T = typename Cfg::T;
  static_assert(Cfg::NP >= 2, "the strided tier needs at least two passes (LDS exchange)");
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem_strided);
  const unsigned f = threadIdx.x % Cfg::FPW;
  const unsigned tid = threadIdx.x / Cfg::FPW;
  const cx<T>* __restrict__ tw = static_cast<const cx<T>*>(a.tw);
  const long long ngroups = (a.total + Cfg::FPW - 1) / Cfg::FPW;
  long long g = blockIdx.x;
  if (g >= ngroups) return;
  cx<T> cur[Cfg::bpt(0)][Cfg::Seq::r[0]];
  cx<T> nxt[Cfg::bpt(0)][Cfg::Seq::r[0]];
  __amdgpu_buffer_rsrc_t rin, rout, rin_n, rout_n;
  bool live, live_n;
  long long c0, c0_n;
  strided_group<Cfg>(a, g, f, &rin, &rout, &live, &c0);
  strided_pass0_load<Cfg, BWD>(rin, a, f, tid, live, cur);
  for (; g < ngroups; g += gridDim.x) {
    strided_pass0_compute<Cfg>(cur, f, tid, lds);
    const long long gn = g + gridDim.x;
    if (gn < ngroups) {
      strided_group<Cfg>(a, gn, f, &rin_n, &rout_n, &live_n, &c0_n);
      strided_pass0_load<Cfg, BWD>(rin_n, a, f, tid, live_n, nxt);
    }
    strided_passes<Cfg, BWD, STW, 1>(rin, rout, a, f, tid, live, c0, lds, tw);
    sfor<0, Cfg::bpt(0)>([&](auto i_) PFA_LAMBDA {
      sfor<0, Cfg::Seq::r[0]>([&](auto t_) PFA_LAMBDA { cur[decltype(i_)::value][decltype(t_)::value] = nxt[decltype(i_)::value][decltype(t_)::value]; });
    });
    rin = rin_n;
    rout = rout_n;
    live = live_n;
    c0 = c0_n;
  }
}

/// LDS bytes of the strided kernel for a wg_cfg (unpadded [element][f] image)
template <typename Cfg>
constexpr size_t strided_lds_bytes() {
  return Cfg::NP > 1 ? size_t(Cfg::N) * Cfg::FPW * sizeof(cx<typename Cfg::T>) : 0;
}

template <typename Cfg, bool BWD, bool STW>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_strided_kernel(const strided_args a) {
  using T = typename Cfg::T;
  static_assert(Cfg::NP >= 2, "the strided tier needs at least two passes (LDS exchange)");
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem_strided);
  const unsigned f = threadIdx.x % Cfg::FPW;
  const unsigned tid = threadIdx.x / Cfg::FPW;
  const cx<T>* __restrict__ tw = static_cast<const cx<T>*>(a.tw);
  const long long ngroups = (a.total + Cfg::FPW - 1) / Cfg::FPW;
  constexpr unsigned ES = sizeof(cx<T>);
  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const long long t0 = g * Cfg::FPW;
    const long long o = t0 / a.inner;
    const long long c0 = t0 - o * a.inner;
    const long long left = a.total - t0;
    const bool live = static_cast<long long>(f) < left;
    const cx<T>* in0 = static_cast<const cx<T>*>(a.in) + o * a.in_dist_outer + c0 * a.in_fdist;
    cx<T>* out0 = static_cast<cx<T>*>(a.out) + o * a.out_dist_outer + c0 * a.out_fdist;
    // ranges: last element of the last FFT of the group (the planner guarantees < 4 GiB)
    const unsigned in_bytes =
        (static_cast<unsigned>(Cfg::FPW - 1) * a.in_fdist + static_cast<unsigned>(Cfg::N - 1) * a.in_stride + 1) * ES;
    const unsigned out_bytes =
        (static_cast<unsigned>(Cfg::FPW - 1) * a.out_fdist + static_cast<unsigned>(Cfg::N - 1) * a.out_stride + 1) * ES;
    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<cx<T>*>(in0), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out0, 0, out_bytes, 0x00020000);
    // no barrier needed here: the last pass ends its LDS reads with a barrier before the next group's first write
    strided_passes<Cfg, BWD, STW, 0>(rin, rout, a, f, tid, live, c0, lds, tw);
  }
}

}  // namespace pfa
