// Sub-group tier with a cross-lane exchange: N = R * R (R = 4, 8, 16), R lanes per FFT, the exchange between the two
// radix-R passes is an R x R transpose ACROSS THE R LANES of the FFT inside one wave64 -- DPP quad permutes for the
// lane-xor 1 and 2 steps, row rotate for xor 8, ds_swizzle for xor 4 -- instead of a round trip through LDS.
//
// Role in the reference: sg_dft / cross-lane transposes of the SUBGROUP level
// (/root/reference/src/portfft/common/subgroup.hpp:141-216, 271-291: sycl::select_from_group shuffles between the
// work-items of a sub-group).  BASELINE.json's north_star names this tier ("wavefront-64 cross-lane shuffles");
// the production path serves these lengths with the LDS-staged Stockham kernel (stockham_wg.hpp, STAGED), and this
// kernel exists so that the choice is a measurement: profiles/r2_notes.md has the A/B.
//
// Data flow per group of FPW FFTs: coalesced copy HBM -> LDS (same as STAGED), pass 0 reads its R inputs from LDS
// (stride R), butterfly, in-wave transpose, twiddle + butterfly, results back to LDS in natural order, coalesced
// copy LDS -> HBM.  The two lanes-of-one-FFT phases need no barrier: an FFT's image is only touched by its own R
// lanes, which sit in one wave.
#pragma once
#include "stockham_wg.hpp"

namespace pfa {

/// value of lane (l ^ K) for a 32-bit register; K in {1, 2, 4, 8}
template <int K>
PFA_DEV int lane_xor_b32(int x) {
  if constexpr (K == 1) {
    return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  } else if constexpr (K == 2) {
    return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  } else if constexpr (K == 8) {
    return __builtin_amdgcn_mov_dpp(x, 0x128, 0xF, 0xF, true);  // row_ror:8 (rows of 16 lanes)
  } else {
    static_assert(K == 4, "lane_xor: K must be 1, 2, 4 or 8");
    return __builtin_amdgcn_ds_swizzle(x, 0x101F);  // bit mode: and 0x1F, or 0, xor 4
  }
}

template <int K, typename T>
PFA_DEV cx<T> lane_xor(cx<T> v) {
  constexpr int W = sizeof(cx<T>) / 4;
  struct words { int w[W]; };
  words a = __builtin_bit_cast(words, v);
  sfor<0, W>([&](auto i_) PFA_LAMBDA { a.w[decltype(i_)::value] = lane_xor_b32<K>(a.w[decltype(i_)::value]); });
  return __builtin_bit_cast(cx<T>, a);
}

/// one step of the transpose: bit K of the lane index and bit K of the register index change places
template <int R, int K, typename T>
PFA_DEV void xlane_stage(cx<T> (&v)[R], unsigned lane) {
  const bool hi = (lane & K) != 0;
  sfor<0, R>([&](auto a_) PFA_LAMBDA {
    constexpr int a = decltype(a_)::value;
    if constexpr ((a & K) == 0) {
      constexpr int b = a | K;
      // value selects only: a conditional store would become a dynamically indexed one and push v[] to scratch
      const cx<T> va = v[a], vb = v[b];
      const cx<T> send = {hi ? va.re : vb.re, hi ? va.im : vb.im};
      const cx<T> recv = lane_xor<K>(send);
      v[a] = {hi ? recv.re : va.re, hi ? recv.im : va.im};
      v[b] = {hi ? vb.re : recv.re, hi ? vb.im : recv.im};
    }
  });
}

/// lane j, register t  <->  lane t, register j  among R consecutive lanes
template <int R, typename T>
PFA_DEV void xlane_transpose(cx<T> (&v)[R], unsigned lane) {
  xlane_stage<R, 1>(v, lane);
  if constexpr (R >= 4) xlane_stage<R, 2>(v, lane);
  if constexpr (R >= 8) xlane_stage<R, 4>(v, lane);
  if constexpr (R >= 16) xlane_stage<R, 8>(v, lane);
}

template <typename Cfg>
constexpr bool xlane_supported() {
  return Cfg::NP == 2 && Cfg::Seq::r[0] == Cfg::Seq::r[1] && Cfg::TPF == Cfg::Seq::r[0] && Cfg::STAGED == 1 &&
         (Cfg::Seq::r[0] == 4 || Cfg::Seq::r[0] == 8 || Cfg::Seq::r[0] == 16) && Cfg::WG % 64 == 0;
}

template <typename Cfg, bool BWD>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_wg_xlane_kernel(
    const cx<typename Cfg::T>* in, cx<typename Cfg::T>* out,
    const cx<typename Cfg::T>* __restrict__ tw, long long nfft, typename Cfg::T scale) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  static_assert(xlane_supported<Cfg>(), "cross-lane kernel: N = R * R with R lanes per FFT, LDS-staged I/O");
  constexpr int R = Seq::r[0];
  constexpr int N = Cfg::N;
  extern __shared__ __attribute__((aligned(16))) char pfa_smem[];
  cx<T>* all = reinterpret_cast<cx<T>*>(pfa_smem);
  const unsigned f = threadIdx.x / R;
  const unsigned j = threadIdx.x % R;
  cx<T>* img = all + f * Cfg::LDS_PER_FFT;
  // the lane's pass-1 twiddles W_N^(t*j) stay in registers for the work-group's lifetime
  cx<T> w[R];
  sfor<1, R>([&](auto t_) PFA_LAMBDA {
    constexpr int t = decltype(t_)::value;
    w[t] = tw[Seq::tw_off(1) + (t - 1) * R + j];
  });
  using IO = packed_io<T, N, Cfg::FPW, Cfg::AUX>;
  constexpr int CH = Cfg::FPW * N;
  constexpr int EPT = (CH + Cfg::WG - 1) / Cfg::WG;
  const long long ngroups = (nfft + Cfg::FPW - 1) / Cfg::FPW;
  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const IO io(in, out, g, nfft);
    sfor<0, EPT>([&](auto k_) PFA_LAMBDA {
      constexpr int k = decltype(k_)::value;
      const unsigned e = threadIdx.x + k * Cfg::WG;
      if (CH % Cfg::WG == 0 || e < CH) {
        cx<T> x = io.load(io.in_elem(e), 0);
        if constexpr (BWD) x.im = -x.im;
        all[(e / N) * Cfg::LDS_PER_FFT + lds_pad<Cfg>(e % N)] = x;
      }
    });
    __syncthreads();
    cx<T> v[R];
    sfor<0, R>([&](auto t_) PFA_LAMBDA {
      constexpr int t = decltype(t_)::value;
      v[t] = img[lds_pad<Cfg>(j + t * R)];
    });
    dft<R>(v);                       // lane j now holds elements j * R + u of the pass-0 output
    xlane_transpose<R>(v, j);        // lane j now holds elements j + t * R: the inputs of its pass-1 butterfly
    sfor<1, R>([&](auto t_) PFA_LAMBDA {
      constexpr int t = decltype(t_)::value;
      v[t] = cmul(v[t], w[t]);
    });
    dft<R>(v);
    sfor<0, R>([&](auto u_) PFA_LAMBDA {
      constexpr int u = decltype(u_)::value;
      cx<T> y = v[u];
      if constexpr (BWD) y.im = -y.im;
      y.re *= scale;
      y.im *= scale;
      img[lds_pad<Cfg>(j + u * R)] = y;  // only this FFT's own R lanes (one wave) touch its image: no barrier needed
    });
    __syncthreads();
    sfor<0, EPT>([&](auto k_) PFA_LAMBDA {
      constexpr int k = decltype(k_)::value;
      const unsigned e = threadIdx.x + k * Cfg::WG;
      if (CH % Cfg::WG == 0 || e < CH) {
        io.store(all[(e / N) * Cfg::LDS_PER_FFT + lds_pad<Cfg>(e % N)], io.out_elem(e), 0);
      }
    });
    __syncthreads();
  }
}

}  // namespace pfa
