// Strided work-group kernel with the first radix taken in registers ("split first radix"): the lanes load the whole
// group, run the radix-R0 pass 0 on their registers, and the R0 residue classes it produces -- element e of the
// Stockham image belongs to class e % R0, and no later pass ever mixes two classes (every later butterfly stride and
// every later Ns is a multiple of R0) -- go through the remaining passes ONE CLASS AT A TIME through an LDS image of
// 1 / R0 of the group, while the other classes wait in registers.
//
// Why: the stage kernels of the four-step tier hold one 128 KiB group per CU (fp64 n = 1024 x 8 columns, fp32 n = 1024
// x 16 columns), so nothing overlaps a work-group's load, pass and store phases on its CU; the same kernel with two
// resident work-groups runs 8-20 % faster (tools/tune_fourstep.hip: fp32 n = 512 x 16 columns at one / two per CU,
// stage A 119 -> 100 us, stage B 92 -> 85 us per 256 MiB).  The CU has 512 KiB of registers against 160 KiB of LDS:
// with R0 = 2 the image of such a group is 64 KiB and two work-groups fit.  Against the half-exchange kernel
// (stockham_strided_hx.hpp: also half the LDS) the exchanges stay whole complex elements with two barriers each, and
// the registers at the peak are the group itself plus one class in flight.
// Cost: one more exchange per element than the 16.8.8 plan (2.8.8.8: the scatter behind pass 0 counts as one).
//
// Same role as stockham_strided_kernel (stockham_strided.hpp; reference: common/global.hpp:135-170 and the
// BATCH_INTERLEAVED dispatcher branches); same twiddle tables (radix_list::tw_off), addressing (strided_args) and
// store modifier.  Requirements (sfr_supported): no ragged pass, interleaved storage.
#pragma once
#include "../../portfft_amd/csrc/stockham_strided.hpp"

namespace pfa {

template <typename Cfg>
constexpr bool sfr_supported() {
  if (Cfg::NP < 2) return false;
  constexpr int R0 = Cfg::Seq::r[0];
  if ((Cfg::N / R0) % Cfg::TPF != 0) return false;
  for (int p = 1; p < Cfg::NP; ++p) {
    const int nb = Cfg::N / Cfg::Seq::r[p];
    if (nb % R0 != 0 || (nb / R0) % Cfg::TPF != 0) return false;
  }
  return true;
}

/// image of one residue class, complex elements
template <typename Cfg>
constexpr size_t sfr_image_elems() {
  return size_t(Cfg::N / Cfg::Seq::r[0]) * Cfg::FPW;
}
/// LDS bytes: the class image, then the TWL twiddle copy (the launch adds the store-modifier tables of STW == 1)
template <typename Cfg>
constexpr size_t strided_sfr_lds_bytes() {
  return (sfr_image_elems<Cfg>() + size_t(Cfg::TWL_ELEMS)) * sizeof(cx<typename Cfg::T>);
}

/// TWL of an SFR kernel: the leading tables that fit behind the class image without costing a resident work-group
template <typename T, typename Seq, int WG, int FPW>
constexpr int auto_twl_sfr() {
  constexpr long long cu_lds = 160 * 1024;
  const long long base = static_cast<long long>(Seq::n / Seq::r[0]) * FPW * static_cast<long long>(sizeof(cx<T>));
  const long long stw = 8 * 1024;  // room kept for the store-modifier tables of a stage A
  const long long before = cu_lds / (base + stw);
  for (int k = Seq::count - 1; k >= 1; --k) {
    const long long extra = static_cast<long long>(Seq::tw_off(k + 1)) * static_cast<long long>(sizeof(cx<T>));
    if (extra <= 16 * 1024 && base + stw + extra <= cu_lds && cu_lds / (base + stw + extra) >= before) return k;
  }
  return 0;
}
template <typename T, typename Seq, int WG, int FPW, int OCC, int AUX>
using sfr_cfg = wg_cfg<T, Seq, WG, FPW, 0, 0, TW_GLOBAL, OCC, AUX, 0, auto_twl_sfr<T, Seq, WG, FPW>()>;

/// pass P >= 1 on residue class U: butterflies j = jj * R0 + U, the class image holds element e at slot e / R0
template <typename Cfg, bool BWD, int STW, int P, int U, typename IO>
PFA_DEV void sfr_pass(const IO& io, const strided_args& a, unsigned f, unsigned tid, bool live, long long c0,
                      cx<typename Cfg::T>* lds, const cx<typename Cfg::T>* __restrict__ tw) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  constexpr int R0 = Seq::r[0];
  constexpr int R = Seq::r[P];
  constexpr int NB = Cfg::N / R;
  constexpr int NBS = NB / R0;  // butterflies of one class
  constexpr int Ns = Seq::ns(P);
  constexpr int BPT = NBS / Cfg::TPF;
  constexpr int FPW = Cfg::FPW;
  constexpr bool last = P == Cfg::NP - 1;
  static_assert(Ns % R0 == 0 && NB % R0 == 0 && NBS % Cfg::TPF == 0, "see sfr_supported()");
  cx<T> v[BPT][R];
  sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned jj = tid + i * Cfg::TPF;
    const cx<T>* p = lds + jj * FPW + f;  // element j + t * NB (j = jj * R0 + U) sits at slot jj + t * NBS
    sfor<0, R>([&](auto t_) PFA_LAMBDA {
      constexpr int t = decltype(t_)::value;
      v[i][t] = p[t * NBS * FPW];
    });
  });
  __syncthreads();
  sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned jj = tid + i * Cfg::TPF;
    const unsigned j = jj * R0 + U;
    const unsigned q = j % Ns;
    sfor<1, R>([&](auto t_) PFA_LAMBDA {
      constexpr int t = decltype(t_)::value;
      cx<T> w;
      if constexpr (P <= Cfg::TWL) {
        w = (lds + sfr_image_elems<Cfg>() + Seq::tw_off(P) + (t - 1) * Ns)[q];
      } else {
        w = (tw + Seq::tw_off(P) + (t - 1) * Ns)[q];
      }
      v[i][t] = cmul(v[i][t], w);
    });
    dft<R>(v[i]);
    const unsigned base = (j / Ns) * (Ns * R) + q;  // element index in the full Stockham image
    if constexpr (last) {
      strided_store_butterfly<Cfg, BWD, STW, R, Ns, IO, R0>(io, a, f, base, live, c0, v[i]);
    } else {
      cx<T>* p = lds + (base / R0) * FPW + f;  // base % R0 == U
      sfor<0, R>([&](auto u_) PFA_LAMBDA {
        constexpr int u = decltype(u_)::value;
        p[u * (Ns / R0) * FPW] = v[i][u];
      });
    }
  });
  if constexpr (!last) __syncthreads();
}

template <typename Cfg, bool BWD, int STW, int P, int U, typename IO>
PFA_DEV void sfr_passes(const IO& io, const strided_args& a, unsigned f, unsigned tid, bool live, long long c0,
                        cx<typename Cfg::T>* lds, const cx<typename Cfg::T>* __restrict__ tw) {
  if constexpr (P < Cfg::NP) {
    sfr_pass<Cfg, BWD, STW, P, U, IO>(io, a, f, tid, live, c0, lds, tw);
    sfr_passes<Cfg, BWD, STW, P + 1, U, IO>(io, a, f, tid, live, c0, lds, tw);
  }
}

template <typename Cfg, bool BWD, int STW>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_strided_sfr_kernel(const strided_args a) {
  using T = typename Cfg::T;
  static_assert(sfr_supported<Cfg>(), "split-first-radix kernel: see sfr_supported()");
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem_strided);
  constexpr int R0 = Cfg::Seq::r[0];
  constexpr int BPT0 = Cfg::bpt(0);
  const unsigned f = threadIdx.x % Cfg::FPW;
  const unsigned tid = threadIdx.x / Cfg::FPW;
  const cx<T>* __restrict__ tw = static_cast<const cx<T>*>(a.tw);
  const long long ngroups = strided_ngroups<Cfg>(a);
  if constexpr (Cfg::TWL > 0) {
    cx<T>* twl = lds + sfr_image_elems<Cfg>();
    for (int i = threadIdx.x; i < Cfg::TWL_ELEMS; i += Cfg::WG) twl[i] = tw[i];
    __syncthreads();
  }
  strided_copy_stw<Cfg, STW, R0>(a);
  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    bool live;
    long long c0;
    const auto io = strided_group<Cfg, 0>(a, g, f, &live, &c0);
    cx<T> v[BPT0][R0];
    strided_pass0_load<Cfg, BWD>(io, a, f, tid, live, v);
    sfor<0, BPT0>([&](auto i_) PFA_LAMBDA { dft<R0>(v[decltype(i_)::value]); });
    sfor<0, R0>([&](auto u_) PFA_LAMBDA {
      constexpr int u = decltype(u_)::value;
      // class u: pass 0 left element j * R0 + u (butterfly j = tid + i * TPF) in v[i][u]; it goes to slot j.  The
      // previous class's last pass has ended its LDS reads with a barrier.
      sfor<0, BPT0>([&](auto i_) PFA_LAMBDA {
        constexpr int i = decltype(i_)::value;
        lds[(tid + i * Cfg::TPF) * Cfg::FPW + f] = v[i][u];
      });
      __syncthreads();
      sfr_passes<Cfg, BWD, STW, 1, u, decltype(io)>(io, a, f, tid, live, c0, lds, tw);
    });
  }
}

}  // namespace pfa
