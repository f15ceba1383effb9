// Fused multi-dimensional transforms: a whole (small) N-D FFT per work-group, one HBM round trip.
//
// Role in the reference: dispatch_dimensions (/root/reference/src/portfft/committed_descriptor_impl.hpp:923-948)
// runs one kernel launch per dimension -- and for the outer dimensions one per (batch, outer index) -- each a full
// pass over HBM.  When the flattened transform fits LDS there is no reason to leave the chip between dimensions:
// the work-group copies its FPW transforms HBM -> LDS with full-line accesses, runs the Stockham passes of every
// dimension LDS -> LDS, and copies the result back.  2-D 64x64 or 3-D 16x16x16 batches then cost 2 x N x sizeof
// bytes of HBM traffic instead of 2 x rank x that.
//
// A pass is described by nd_pass<R, NS, L, S, TWOFF>: radix R, Stockham stride NS (product of the earlier radices
// of the same dimension), dimension length L, memory stride S of that dimension (product of the later dimensions'
// lengths) and the offset of its twiddles.  Butterfly b of a pass decomposes as (o_lo, jb, o_hi) =
// (b % S, (b / S) % (L / R), b / (S * L / R)): lanes run over the contiguous index first, so every dimension but the
// last one reads and writes lane-contiguous LDS addresses; the last dimension (S = 1) has the access pattern of the
// 1-D kernel (stockham_wg.hpp) and uses the same padding.
//
// Only instantiated at commit time (jit.cpp): the shapes are too many to pre-compile.
#pragma once
#include "stockham_wg.hpp"

namespace pfa {

template <int R_, int NS_, int L_, int S_, int TWOFF_>
struct nd_pass {
  static constexpr int R = R_;
  static constexpr int NS = NS_;
  static constexpr int L = L_;
  static constexpr int S = S_;
  static constexpr int TWOFF = TWOFF_;
};

template <typename T_, int NTOT_, int WG_, int FPW_, int PADS_, int PADW_, int OCC_, int AUX_, typename... Passes>
struct nd_cfg {
  using T = T_;
  static constexpr int N = NTOT_;
  static constexpr int WG = WG_;
  static constexpr int FPW = FPW_;
  static constexpr int TPF = WG_ / FPW_;
  static constexpr int PADS = PADS_;
  static constexpr int PADW = PADW_;
  static constexpr int OCC = OCC_;
  static constexpr int AUX = AUX_;
  static constexpr int pad(int i) { return PADS_ == 0 ? i : i + ((i / PADS_) * PADW_); }
  static constexpr int LDS_PER_FFT = pad(N - 1) + 1 + (PADS_ == 0 ? 0 : PADW_);
  static constexpr size_t LDS_BYTES = size_t(LDS_PER_FFT) * FPW_ * sizeof(cx<T_>);
  template <typename F>
  static PFA_DEV void for_each_pass(F&& f) {
    (f(Passes{}), ...);
  }
};

template <typename Cfg, typename P>
PFA_DEV void nd_run_pass(cx<typename Cfg::T>* lds, int tid, const cx<typename Cfg::T>* __restrict__ tw) {
  using T = typename Cfg::T;
  constexpr int R = P::R, NS = P::NS, L = P::L, S = P::S;
  constexpr int LB = L / R;            // butterflies along the dimension
  constexpr int NB = Cfg::N / R;       // butterflies per transform
  constexpr int BPT = (NB + Cfg::TPF - 1) / Cfg::TPF;
  constexpr bool ragged = (NB % Cfg::TPF) != 0;
  cx<T> v[BPT][R];
  sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned b = tid + i * Cfg::TPF;
    if (!ragged || b < NB) {
      const unsigned o_lo = b % S, jb = (b / S) % LB, o_hi = b / (S * LB);
      const unsigned base = o_hi * (L * S) + o_lo;
      sfor<0, R>([&](auto t_) PFA_LAMBDA {
        constexpr int t = decltype(t_)::value;
        v[i][t] = lds[lds_pad<Cfg>(base + (jb + t * LB) * S)];
      });
    }
  });
  __syncthreads();
  sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned b = tid + i * Cfg::TPF;
    if (!ragged || b < NB) {
      const unsigned o_lo = b % S, jb = (b / S) % LB, o_hi = b / (S * LB);
      const unsigned base = o_hi * (L * S) + o_lo;
      const unsigned q = jb % NS;
      if constexpr (NS > 1) {
        sfor<1, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          v[i][t] = cmul(v[i][t], (tw + P::TWOFF + (t - 1) * NS)[q]);
        });
      }
      dft<R>(v[i]);
      const unsigned k0 = (jb / NS) * (NS * R) + q;
      sfor<0, R>([&](auto u_) PFA_LAMBDA {
        constexpr int u = decltype(u_)::value;
        lds[lds_pad<Cfg>(base + (k0 + u * NS) * S)] = v[i][u];
      });
    }
  });
  __syncthreads();
}

template <typename Cfg, bool BWD, typename MakeIO>
PFA_DEV void stockham_nd_body(MakeIO&& make_io, const cx<typename Cfg::T>* __restrict__ tw, long long nfft,
                              typename Cfg::T scale) {
  using T = typename Cfg::T;
  extern __shared__ __attribute__((aligned(16))) char pfa_smem[];
  cx<T>* all = reinterpret_cast<cx<T>*>(pfa_smem);
  const int f = threadIdx.x / Cfg::TPF;
  const int tid = threadIdx.x % Cfg::TPF;
  cx<T>* lds = all + f * Cfg::LDS_PER_FFT;
  constexpr int CH = Cfg::FPW * Cfg::N;
  constexpr int EPT = (CH + Cfg::WG - 1) / Cfg::WG;
  const long long ngroups = (nfft + Cfg::FPW - 1) / Cfg::FPW;
  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const auto io = make_io(g);
    sfor<0, EPT>([&](auto k_) PFA_LAMBDA {
      constexpr int k = decltype(k_)::value;
      const unsigned e = threadIdx.x + k * Cfg::WG;
      if (CH % Cfg::WG == 0 || e < CH) {
        cx<T> x = io.load(io.in_elem(e), 0);
        if constexpr (BWD) x.im = -x.im;
        all[(e / Cfg::N) * Cfg::LDS_PER_FFT + lds_pad<Cfg>(e % Cfg::N)] = x;
      }
    });
    __syncthreads();
    const cx<T>* twp = tw;
    asm volatile("" : "+s"(twp));  // keep the table reads inside the loop (L1/L2 hits), not pinned in VGPRs
    Cfg::for_each_pass([&](auto pass) PFA_LAMBDA { nd_run_pass<Cfg, decltype(pass)>(lds, tid, twp); });
    sfor<0, EPT>([&](auto k_) PFA_LAMBDA {
      constexpr int k = decltype(k_)::value;
      const unsigned e = threadIdx.x + k * Cfg::WG;
      if (CH % Cfg::WG == 0 || e < CH) {
        cx<T> y = all[(e / Cfg::N) * Cfg::LDS_PER_FFT + lds_pad<Cfg>(e % Cfg::N)];
        if constexpr (BWD) y.im = -y.im;
        y.re *= scale;
        y.im *= scale;
        io.store(y, io.out_elem(e), 0);
      }
    });
    __syncthreads();  // the next group's copy-in overwrites the images
  }
}

/// interleaved complex; same launch signature as stockham_wg_kernel
template <typename Cfg, bool BWD>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_nd_kernel(const cx<typename Cfg::T>* in,
                                                                        cx<typename Cfg::T>* out,
                                                                        const cx<typename Cfg::T>* __restrict__ tw,
                                                                        long long nfft, typename Cfg::T scale) {
  using T = typename Cfg::T;
  stockham_nd_body<Cfg, BWD>(
      [&](long long g) PFA_LAMBDA { return packed_io<T, Cfg::N, Cfg::FPW, Cfg::AUX>(in, out, g, nfft); }, tw, nfft,
      scale);
}

/// SPLIT_COMPLEX storage; same launch signature as stockham_wg_split_kernel
template <typename Cfg, bool BWD>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_nd_split_kernel(
    const typename Cfg::T* in_re, const typename Cfg::T* in_im,
    typename Cfg::T* out_re, typename Cfg::T* out_im,
    const cx<typename Cfg::T>* __restrict__ tw, long long nfft, typename Cfg::T scale) {
  using T = typename Cfg::T;
  stockham_nd_body<Cfg, BWD>(
      [&](long long g) PFA_LAMBDA {
        return packed_split_io<T, Cfg::N, Cfg::FPW, Cfg::AUX>(in_re, in_im, out_re, out_im, g, nfft);
      },
      tw, nfft, scale);
}

}  // namespace pfa
