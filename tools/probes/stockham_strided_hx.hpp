// Strided work-group kernel with a half-size LDS image ("half exchange"): the lanes keep the whole group in
// registers and exchange the real parts, then the imaginary parts, through one scalar image [element][f].
//
// Why: the column passes of the four-step tier are bound by their HBM access shape (FPW adjacent columns x 16 B per
// fp64 element = 128-byte segments at a row pitch) and by running one 128 KiB work-group per CU (fp64 n = 1024 x 8
// columns), so that nothing overlaps a work-group's load, compute and store phases.  The register file of a CU is
// 512 KiB, its LDS 160 KiB: with the data resident in registers and only one component in LDS at a time the same
// group needs half the LDS -- two work-groups per CU at 8 columns, or 16 columns (256-byte segments) in one.
// Cost: four barriers per exchange instead of two, 1.5x the data registers at the peak.
//
// Same role as stockham_strided_kernel (stockham_strided.hpp; reference: common/global.hpp:135-170 and the
// BATCH_INTERLEAVED dispatcher branches); same twiddle tables, addressing (strided_args) and store modifier.
// Requirements: every pass divides evenly over the lanes (no ragged pass), interleaved storage.
#pragma once
#include "../../portfft_amd/csrc/stockham_strided.hpp"

namespace pfa {

template <typename Cfg>
constexpr bool hx_supported() {
  if (Cfg::NP < 2) return false;
  for (int p = 0; p < Cfg::NP; ++p) {
    if ((Cfg::N / Cfg::Seq::r[p]) % Cfg::TPF != 0) return false;
  }
  return true;
}

/// LDS bytes: one scalar image, then the TWL twiddle copy (the launch adds the store-modifier tables of STW == 1)
template <typename Cfg>
constexpr size_t strided_hx_lds_bytes() {
  static_assert((Cfg::N * Cfg::FPW) % 2 == 0, "the scalar image must end on a complex element");
  return size_t(Cfg::N) * Cfg::FPW * sizeof(typename Cfg::T) + size_t(Cfg::TWL_ELEMS) * sizeof(cx<typename Cfg::T>);
}

/// twiddle + butterfly of pass P on the lane's registers
template <typename Cfg, int P>
PFA_DEV void hx_compute(cx<typename Cfg::T> (&v)[Cfg::bpt(P)][Cfg::Seq::r[P]], unsigned tid,
                        const cx<typename Cfg::T>* twl, const cx<typename Cfg::T>* __restrict__ tw) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  constexpr int R = Seq::r[P];
  constexpr int Ns = Seq::ns(P);
  sfor<0, Cfg::bpt(P)>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned q = (tid + i * Cfg::TPF) % Ns;
    if constexpr (P != 0) {
      sfor<1, R>([&](auto t_) PFA_LAMBDA {
        constexpr int t = decltype(t_)::value;
        cx<T> w;
        if constexpr (P <= Cfg::TWL) {
          w = (twl + Seq::tw_off(P) + (t - 1) * Ns)[q];
        } else {
          w = (tw + Seq::tw_off(P) + (t - 1) * Ns)[q];
        }
        v[i][t] = cmul(v[i][t], w);
      });
    }
    dft<R>(v[i]);
  });
}

/// exchange between pass P and pass P + 1: component by component through the scalar image
template <typename Cfg, int P>
PFA_DEV void hx_exchange(cx<typename Cfg::T> (&v)[Cfg::bpt(P)][Cfg::Seq::r[P]],
                         cx<typename Cfg::T> (&n)[Cfg::bpt(P + 1)][Cfg::Seq::r[P + 1]], unsigned f, unsigned tid,
                         typename Cfg::T* img) {
  using Seq = typename Cfg::Seq;
  constexpr int R = Seq::r[P], Ns = Seq::ns(P);
  constexpr int R1 = Seq::r[P + 1], NB1 = Cfg::N / R1;
  constexpr int FPW = Cfg::FPW;
  sfor<0, 2>([&](auto c_) PFA_LAMBDA {
    constexpr int c = decltype(c_)::value;
    sfor<0, Cfg::bpt(P)>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      const unsigned j = tid + i * Cfg::TPF;
      const unsigned base = (j / Ns) * (Ns * R) + j % Ns;
      typename Cfg::T* p = img + base * FPW + f;
      sfor<0, R>([&](auto u_) PFA_LAMBDA {
        constexpr int u = decltype(u_)::value;
        p[u * Ns * FPW] = c == 0 ? v[i][u].re : v[i][u].im;
      });
    });
    __syncthreads();
    sfor<0, Cfg::bpt(P + 1)>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      const unsigned j = tid + i * Cfg::TPF;
      const typename Cfg::T* p = img + j * FPW + f;
      sfor<0, R1>([&](auto t_) PFA_LAMBDA {
        constexpr int t = decltype(t_)::value;
        if constexpr (c == 0) {
          n[i][t].re = p[t * NB1 * FPW];
        } else {
          n[i][t].im = p[t * NB1 * FPW];
        }
      });
    });
    __syncthreads();
  });
}

template <typename Cfg, bool BWD, int STW, int P, typename IO>
PFA_DEV void hx_passes(cx<typename Cfg::T> (&v)[Cfg::bpt(P)][Cfg::Seq::r[P]], const IO& io, const strided_args& a,
                       unsigned f, unsigned tid, bool live, long long c0, typename Cfg::T* img,
                       const cx<typename Cfg::T>* twl, const cx<typename Cfg::T>* __restrict__ tw) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  hx_compute<Cfg, P>(v, tid, twl, tw);
  if constexpr (P == Cfg::NP - 1) {
    constexpr int R = Seq::r[P], Ns = Seq::ns(P);
    sfor<0, Cfg::bpt(P)>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      const unsigned j = tid + i * Cfg::TPF;
      const unsigned base = (j / Ns) * (Ns * R) + j % Ns;
      strided_store_butterfly<Cfg, BWD, STW, R, Ns, IO, 2>(io, a, f, base, live, c0, v[i]);
    });
  } else {
    cx<T> n[Cfg::bpt(P + 1)][Seq::r[P + 1]];
    hx_exchange<Cfg, P>(v, n, f, tid, img);
    hx_passes<Cfg, BWD, STW, P + 1>(n, io, a, f, tid, live, c0, img, twl, tw);
  }
}

template <typename Cfg, bool BWD, int STW>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_strided_hx_kernel(const strided_args a) {
  using T = typename Cfg::T;
  static_assert(hx_supported<Cfg>(), "half-exchange kernel: every pass must divide evenly over the lanes");
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  T* img = reinterpret_cast<T*>(pfa_smem_strided);
  cx<T>* twl = reinterpret_cast<cx<T>*>(img + Cfg::N * Cfg::FPW);
  const unsigned f = threadIdx.x % Cfg::FPW;
  const unsigned tid = threadIdx.x / Cfg::FPW;
  const cx<T>* __restrict__ tw = static_cast<const cx<T>*>(a.tw);
  const long long ngroups = strided_ngroups<Cfg>(a);
  if constexpr (Cfg::TWL > 0) {
    for (int i = threadIdx.x; i < Cfg::TWL_ELEMS; i += Cfg::WG) twl[i] = tw[i];
    __syncthreads();
  }
  strided_copy_stw<Cfg, STW, 2>(a);
  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    bool live;
    long long c0;
    const auto io = strided_group<Cfg, 0>(a, g, f, &live, &c0);
    cx<T> v[Cfg::bpt(0)][Cfg::Seq::r[0]];
    strided_pass0_load<Cfg, BWD>(io, a, f, tid, live, v);
    hx_passes<Cfg, BWD, STW, 0>(v, io, a, f, tid, live, c0, img, twl, tw);
  }
}

}  // namespace pfa
