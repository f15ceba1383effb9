// Strided work-group tier, register-resident form: the group (FPW transforms side by side) stays in the REGISTERS of its
// work-group, the exchange between two passes goes through an LDS image of HALF the group in two rounds.
//
// Role in the reference: the sub-kernels of the GLOBAL level -- one kernel shape for any factor of any factor list, with
// the inter-factor twiddles as a store modifier (/root/reference/src/portfft/common/global.hpp:135-170,
// dispatcher/global_dispatcher.hpp:120-167; the reference's regression size 68640, test/unit_test/
// instantiate_fft_tests.hpp:153-157) -- and the BATCH_INTERLEAVED work-group branch (workgroup_dispatcher.hpp:148-229).
//
// Design (ours, MI355X-specific).  stockham_strided_kernel keeps the whole group in LDS: a runtime-specialised four-step
// stage of 600 ... 1000 points x 16 columns needs 80 ... 128 KiB, so ONE work-group of 10 ... 13 waves sits on a CU and its
// three phases (HBM loads | passes through LDS behind barriers | HBM stores) overlap with nothing (profiles/r4_sq_counters.txt:
// 2.6 waves per SIMD, VALU issue 0.25, waits 0.46).  What fixed exactly that shape for packed transforms in round 5
// (stockham_wg_hx.hpp, profiles/r5_sq_counters_pairs.txt) is applied to the strided tier here: lane (f, tid) holds its
// butterflies' values for the whole transform; between pass p (radix R, stride Ns) and pass p + 1 (radix R1, NB1 = N / R1
// butterflies) the elements of every transform split at H = ceil(R1 / 2) * NB1 -- round 0 moves elements [0, H), round 1
// [H, N) through the image [element - h * H][f] (f fastest: every LDS access lane-contiguous, as in the LDS-resident
// kernel).  Half the LDS per work-group: TWO (or three) work-groups share a CU and cover each other's HBM phases.
// Ragged passes are straight-line code (see stockham_wg_hx.hpp): a slot past the last butterfly loads out of the buffer's
// range (zeros), computes on whatever it holds, re-reads the last butterfly's legs from LDS and stores out of range;
// only the LDS writes are predicated.
//
// Same launch-time arguments (strided_args), group addressing (strided_group), store modifier (stw_apply: STW 1 LDS
// tables / 2 global tables) and storage forms (SPLIT 0 ... 3) as stockham_strided_kernel: the planner swaps the kernel, not
// the stage.  Not here: the row-staged and tiled-input forms (they belong to lengths whose image is small anyway).
#pragma once
#include "stockham_strided.hpp"
#include "stockham_wg_hx.hpp"

namespace pfa {

template <typename Cfg>
constexpr bool strided_hx_supported() {
  return Cfg::NP >= 2 && Cfg::STAGED == 0 && Cfg::PADS == 0;
}

/// elements PER TRANSFORM of the image: the largest first half of any exchange
template <typename Cfg>
constexpr int strided_hx_image_elems() {
  int h = 0;
  for (int p = 0; p + 1 < Cfg::NP; ++p) h = wg_hx_split<Cfg>(p) > h ? wg_hx_split<Cfg>(p) : h;
  return h;
}
/// LDS bytes of the kernel itself: the half image of the group, then the TWL twiddle copy (the store-modifier tables sit
/// behind that: strided_args::stw_lds_off, set by the launch)
template <typename Cfg>
constexpr size_t strided_hx_lds_bytes() {
  return (size_t(strided_hx_image_elems<Cfg>()) * Cfg::FPW + Cfg::TWL_ELEMS) * sizeof(cx<typename Cfg::T>);
}

/// pass 0's loads; dead lanes (transform beyond the end of its outer index, slot past the last butterfly) read out of
/// the buffer's range: zeros, unpredicated
template <typename Cfg, bool BWD, typename IO>
PFA_DEV void shx_load(const IO& io, const strided_args& a, unsigned f, unsigned tid, bool live,
                      cx<typename Cfg::T> (&v)[Cfg::bpt(0)][Cfg::Seq::r[0]]) {
  using T = typename Cfg::T;
  constexpr int R = Cfg::Seq::r[0], NB = Cfg::N / R;
  constexpr unsigned ES = IO::ES_IN;
  const unsigned tsh = static_cast<unsigned>(a.in_tile_shift);
  sfor<0, Cfg::bpt(0)>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    constexpr bool none = i * Cfg::TPF >= NB, all = (i + 1) * Cfg::TPF <= NB;
    if constexpr (!none) {
      const bool ok = live && (all || j < static_cast<unsigned>(NB));
      const unsigned voff =
          ok ? (f * a.in_fdist + (j >> tsh) * a.in_stride + (j & ((1u << tsh) - 1u))) * ES : 0xFFFFFFF0u;
      sfor<0, R>([&](auto t_) PFA_LAMBDA {
        constexpr int t = decltype(t_)::value;
        cx<T> x = io.load(voff, static_cast<typename IO::off_t>(static_cast<unsigned>(t * NB) >> tsh) * a.in_stride * ES);
        if constexpr (BWD) x.im = -x.im;
        v[i][t] = x;
      });
    }
  });
}

/// exchange between pass P and pass P + 1 through the half image, in two rounds (hxw_exchange with FPW transforms side
/// by side: element e of transform f at (e - h * H) * FPW + f)
template <typename Cfg, int P>
PFA_DEV void shx_exchange(cx<typename Cfg::T> (&v)[Cfg::bpt(P)][Cfg::Seq::r[P]],
                          cx<typename Cfg::T> (&n)[Cfg::bpt(P + 1)][Cfg::Seq::r[P + 1]], unsigned f, unsigned tid,
                          cx<typename Cfg::T>* img) {
  using Seq = typename Cfg::Seq;
  constexpr int FPW = Cfg::FPW;
  constexpr int R = Seq::r[P], Ns = Seq::ns(P), NB = Cfg::N / R, BPT = Cfg::bpt(P);
  constexpr int R1 = Seq::r[P + 1], NB1 = Cfg::N / R1, BPT1 = Cfg::bpt(P + 1);
  constexpr int HL = (R1 + 1) / 2;        // legs of a pass-(P + 1) butterfly read in round 0
  constexpr int H = wg_hx_split<Cfg>(P);  // = HL * NB1, a multiple of Ns * R
  constexpr int J0 = H / R;               // butterflies of pass P that write in round 0
  static_assert(H % (Ns * R) == 0 && J0 * R == H, "the split point lies between two butterflies' outputs");
  sfor<0, 2>([&](auto h_) PFA_LAMBDA {
    constexpr int h = decltype(h_)::value;
    sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      const unsigned j = tid + i * Cfg::TPF;
      constexpr int lo = h == 0 ? 0 : J0, hi = h == 0 ? J0 : NB;
      constexpr bool none = (i + 1) * Cfg::TPF <= lo || i * Cfg::TPF >= hi;
      constexpr bool all = i * Cfg::TPF >= lo && (i + 1) * Cfg::TPF <= hi;
      if constexpr (!none) {
        if (all || (j >= static_cast<unsigned>(lo) && j < static_cast<unsigned>(hi))) {
          const unsigned base = (j / Ns) * (Ns * R) + j % Ns - h * H;
          cx<typename Cfg::T>* p = img + base * FPW + f;
          sfor<0, R>([&](auto u_) PFA_LAMBDA {
            constexpr int u = decltype(u_)::value;
            p[u * Ns * FPW] = v[i][u];
          });
        }
      }
    });
    __syncthreads();
    sfor<0, BPT1>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      constexpr bool none1 = i * Cfg::TPF >= NB1, all1 = (i + 1) * Cfg::TPF <= NB1;
      if constexpr (!none1) {
        unsigned j = tid + i * Cfg::TPF;
        if constexpr (!all1) j = j < static_cast<unsigned>(NB1) ? j : static_cast<unsigned>(NB1 - 1);
        constexpr int t0 = h == 0 ? 0 : HL, t1 = h == 0 ? HL : R1;
        const cx<typename Cfg::T>* p = img + j * FPW + f;
        sfor<t0, t1>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          n[i][t] = p[(t - t0) * NB1 * FPW];
        });
      }
    });
    __syncthreads();
  });
}

template <typename Cfg, bool BWD, int STW, int P, typename IO>
PFA_DEV void shx_passes(cx<typename Cfg::T> (&v)[Cfg::bpt(P)][Cfg::Seq::r[P]], const IO& io, const strided_args& a,
                        unsigned f, unsigned tid, bool live, long long c0, cx<typename Cfg::T>* img,
                        const cx<typename Cfg::T>* twl, const cx<typename Cfg::T>* __restrict__ tw) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  constexpr int R = Seq::r[P], Ns = Seq::ns(P), NB = Cfg::N / R;
  sfor<0, Cfg::bpt(P)>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    // (no `j < NB` around the arithmetic of a ragged pass: the idle lanes compute on whatever their registers hold --
    //  their table addresses are valid, nothing of theirs is written)
    if constexpr (P != 0) {
      const unsigned q = (tid + i * Cfg::TPF) % Ns;
      const cx<T>* t0 = (P <= Cfg::TWL ? twl : tw) + Seq::tw_off(P) + q;
      sfor<1, R>([&](auto t_) PFA_LAMBDA {
        constexpr int t = decltype(t_)::value;
        v[i][t] = cmul(v[i][t], t0[(t - 1) * Ns]);
      });
    }
    dft<R>(v[i]);
#ifdef PFA_SHX_SCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
  });
  if constexpr (P == Cfg::NP - 1) {
    sfor<0, Cfg::bpt(P)>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      constexpr bool none_s = i * Cfg::TPF >= NB, all_s = (i + 1) * Cfg::TPF <= NB;
      if constexpr (!none_s) {
        const unsigned j = tid + i * Cfg::TPF;
        // (the store modifier's table indices come from `base`: a slot past the last butterfly takes the last one's)
        const unsigned jc = all_s ? j : (j < static_cast<unsigned>(NB) ? j : static_cast<unsigned>(NB - 1));
        const unsigned base = (jc / Ns) * (Ns * R) + jc % Ns;
        const bool ok = live && (all_s || j < static_cast<unsigned>(NB));
        strided_store_butterfly<Cfg, BWD, STW, R, Ns>(io, a, f, base, ok, c0, v[i]);
      }
    });
  } else {
    cx<T> n[Cfg::bpt(P + 1)][Seq::r[P + 1]];
    shx_exchange<Cfg, P>(v, n, f, tid, img);
    shx_passes<Cfg, BWD, STW, P + 1>(n, io, a, f, tid, live, c0, img, twl, tw);
  }
}

template <typename Cfg, bool BWD, int STW, int SPLIT = 0, bool BIG = false>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_strided_hx_kernel(const strided_args a) {
  using T = typename Cfg::T;
  static_assert(strided_hx_supported<Cfg>(), "see strided_hx_supported()");
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  cx<T>* img = reinterpret_cast<cx<T>*>(pfa_smem_strided);
  cx<T>* twl = img + strided_hx_image_elems<Cfg>() * Cfg::FPW;
  const unsigned f = threadIdx.x % Cfg::FPW;
  const unsigned tid = threadIdx.x / Cfg::FPW;
  const cx<T>* __restrict__ tw = static_cast<const cx<T>*>(a.tw);
  const long long ngroups = strided_ngroups<Cfg>(a);
  if constexpr (Cfg::TWL > 0) {
    for (int i = threadIdx.x; i < Cfg::TWL_ELEMS; i += Cfg::WG) twl[i] = tw[i];
    __syncthreads();
  }
  strided_copy_stw<Cfg, STW>(a);  // (strided_args::stw_lds_off: behind the TWL copy, set by the launch)
  strided_group_walk(a, ngroups, [&](long long g) PFA_LAMBDA {
    bool live;
    long long c0;
    const auto io = strided_group<Cfg, SPLIT, BIG>(a, g, f, &live, &c0);
    cx<T> v[Cfg::bpt(0)][Cfg::Seq::r[0]];
    shx_load<Cfg, BWD>(io, a, f, tid, live, v);
    const cx<T>* twp = tw;
    asm volatile("" : "+s"(twp));  // keep the table reads inside the loop (see stockham_wg_body)
    // (no barrier between two groups: the last exchange ends with one behind its reads)
    shx_passes<Cfg, BWD, STW, 0>(v, io, a, f, tid, live, c0, img, twl, twp);
  });
}

}  // namespace pfa
