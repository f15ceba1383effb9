// Work-group tier, register-resident form: ONE HBM pass for transforms that do not fit the CU's LDS.
//
// Role in the reference: the sizes where portFFT leaves its WORKGROUP level for the GLOBAL one -- fp32 N = 32768 is the
// first GlobalTest size, fp64 8192 / 16384 are "WorkgroupOrGlobal" (/root/reference/test/unit_test/
// instantiate_fft_tests.hpp:140-151; the level rule: src/portfft/committed_descriptor_impl.hpp:210-313) -- where the
// reference takes one kernel per factor through global memory (dispatcher/global_dispatcher.hpp:343-408) and the
// four-step plans of this library move every byte through HBM twice.
//
// Design (ours, MI355X-specific): a CU has 512 KiB of vector registers beside its 160 KiB of LDS.  The transform
// (256 KiB: fp32 2^15, fp64 2^14) stays in the REGISTERS of one 512-lane work-group (64 fp32 complex values per lane)
// for all of its Stockham passes; only the exchange between two passes goes through LDS, and it goes in two rounds
// through an image of HALF the transform: round h moves the elements [h * N/2, (h + 1) * N/2) -- on the writing
// side the butterflies j with (j < NB / 2) == (h == 0) (every output of butterfly j lies in one half, because all
// later radices are even), on the reading side the butterfly legs t with (t < R / 2) == (h == 0).  The registers a
// lane frees by writing are the ones it reads into, so the peak is the transform itself plus a butterfly's temporaries.
// 64-bit (fp64: 128-bit) LDS accesses at their full rate, four barriers per exchange instead of two.
// Pass 0 reads HBM directly, the last pass writes it directly, both lane-contiguous -- 1.0 x the algorithmic traffic.
//
// Twiddles: with 128 data registers per lane there is no room to keep them (the N = 16384 kernel holds 77 complex
// values per lane).  The tables of the leading passes are copied to LDS behind the half image (wg_cfg TWL); the long
// table of the last pass stays in L2 and only every B-th power and the first B - 1 powers are loaded
// (W^(q (B a + b)) = W^(q B a) W^(q b): 10 coalesced loads + 21 multiplies instead of 31 loads for radix 32, each
// twiddle the product of two correctly rounded table values).
#pragma once
#include "stockham_wg.hpp"

namespace pfa {

/// Is the configuration one this kernel can run?  One transform per work-group, no ragged pass (a pass with one
/// butterfly per lane splits its writers by lane halves, which must be whole waves), all radices but the first even.
template <typename Cfg>
constexpr bool wg_hx_supported() {
  if (Cfg::NP < 2 || Cfg::FPW != 1 || Cfg::STAGED != 0 || Cfg::N % 2 != 0) return false;
  for (int p = 0; p < Cfg::NP; ++p) {
    const int nb = Cfg::N / Cfg::Seq::r[p];
    if (nb % Cfg::TPF != 0) return false;
    const int bpt = nb / Cfg::TPF;
    if (bpt % 2 != 0 && !(bpt == 1 && Cfg::TPF % 128 == 0)) return false;
    if (p >= 1 && Cfg::Seq::r[p] % 2 != 0) return false;
  }
  return true;
}

/// elements of the half image (padded like wg_cfg's full image), and the kernel's LDS bytes
template <typename Cfg>
constexpr int wg_hx_image_elems() {
  return Cfg::pad(Cfg::N / 2 - 1) + 1 + (Cfg::PADS == 0 ? 0 : Cfg::PADW);
}
template <typename Cfg>
constexpr size_t wg_hx_lds_bytes() {
  return size_t(wg_hx_image_elems<Cfg>() + Cfg::TWL_ELEMS) * sizeof(cx<typename Cfg::T>);
}

/// the twiddles W^(q t), t = 1 .. R - 1, of one butterfly of pass P, applied to its inputs
template <typename Cfg, int P>
PFA_DEV void hxw_twiddle(cx<typename Cfg::T> (&v)[Cfg::Seq::r[P]], unsigned q, const cx<typename Cfg::T>* twl,
                         const cx<typename Cfg::T>* __restrict__ tw) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  constexpr int R = Seq::r[P];
  constexpr int Ns = Seq::ns(P);
  if constexpr (P <= Cfg::TWL) {
    const cx<T>* t0 = twl + Seq::tw_off(P) + q;
    sfor<1, R>([&](auto t_) PFA_LAMBDA {
      constexpr int t = decltype(t_)::value;
      v[t] = cmul(v[t], t0[(t - 1) * Ns]);
    });
  } else {
    const cx<T>* t0 = tw + Seq::tw_off(P) + q;
    constexpr int B = R >= 32 ? 8 : (R >= 8 ? 4 : R);  // t = B a + b
    cx<T> lo[B];                                       // W^(q b), b = 1 .. B - 1
    sfor<1, B>([&](auto b_) PFA_LAMBDA {
      constexpr int b = decltype(b_)::value;
      lo[b] = t0[(b - 1) * Ns];
    });
    sfor<1, B>([&](auto b_) PFA_LAMBDA {
      constexpr int b = decltype(b_)::value;
      if constexpr (b < R) v[b] = cmul(v[b], lo[b]);
    });
    sfor<1, (R + B - 1) / B>([&](auto a_) PFA_LAMBDA {
      constexpr int a = decltype(a_)::value;
      const cx<T> hi = t0[(B * a - 1) * Ns];  // W^(q B a)
      v[B * a] = cmul(v[B * a], hi);
      sfor<1, B>([&](auto b_) PFA_LAMBDA {
        constexpr int b = decltype(b_)::value;
        if constexpr (B * a + b < R) v[B * a + b] = cmul(v[B * a + b], cmul(hi, lo[b]));
      });
    });
  }
}

/// exchange between pass P and pass P + 1 through the half image, in two rounds
template <typename Cfg, int P>
PFA_DEV void hxw_exchange(cx<typename Cfg::T> (&v)[Cfg::bpt(P)][Cfg::Seq::r[P]],
                          cx<typename Cfg::T> (&n)[Cfg::bpt(P + 1)][Cfg::Seq::r[P + 1]], unsigned tid,
                          cx<typename Cfg::T>* img) {
  using Seq = typename Cfg::Seq;
  constexpr int R = Seq::r[P], Ns = Seq::ns(P), NB = Cfg::N / R, BPT = Cfg::bpt(P);
  constexpr int R1 = Seq::r[P + 1], NB1 = Cfg::N / R1, BPT1 = Cfg::bpt(P + 1);
  constexpr int HALF = Cfg::N / 2;
  sfor<0, 2>([&](auto h_) PFA_LAMBDA {
    constexpr int h = decltype(h_)::value;
    sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      const unsigned j = tid + i * Cfg::TPF;
      // (BPT even: butterfly i of every lane lies in half i / (BPT / 2); BPT == 1: the lanes split, whole waves each)
      const bool mine = BPT > 1 ? (i / ((BPT + 1) / 2) == h) : ((j < NB / 2) == (h == 0));
      if (mine) {
        const unsigned base = (j / Ns) * (Ns * R) + j % Ns - h * HALF;
        if constexpr (pad_is_linear<Cfg>(Ns, R, Ns * R)) {
          cx<typename Cfg::T>* p = img + lds_pad<Cfg>(base);
          sfor<0, R>([&](auto u_) PFA_LAMBDA {
            constexpr int u = decltype(u_)::value;
            p[u * pad_step<Cfg>(Ns)] = v[i][u];
          });
        } else {
          sfor<0, R>([&](auto u_) PFA_LAMBDA {
            constexpr int u = decltype(u_)::value;
            img[lds_pad<Cfg>(base + u * Ns)] = v[i][u];
          });
        }
      }
    });
    __syncthreads();
    sfor<0, BPT1>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      const unsigned j = tid + i * Cfg::TPF;
      if constexpr (pad_is_linear<Cfg>(NB1, R1, 1)) {
        const cx<typename Cfg::T>* p = img + lds_pad<Cfg>(j);
        sfor<h * (R1 / 2), (h + 1) * (R1 / 2)>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          n[i][t] = p[(t - h * (R1 / 2)) * pad_step<Cfg>(NB1)];
        });
      } else {
        sfor<h * (R1 / 2), (h + 1) * (R1 / 2)>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          n[i][t] = img[lds_pad<Cfg>(j + t * NB1 - h * HALF)];
        });
      }
    });
    __syncthreads();
  });
}

template <typename Cfg, bool BWD, int P, typename IO>
PFA_DEV void hxw_passes(cx<typename Cfg::T> (&v)[Cfg::bpt(P)][Cfg::Seq::r[P]], const IO& io, unsigned tid,
                        cx<typename Cfg::T>* img, const cx<typename Cfg::T>* twl,
                        const cx<typename Cfg::T>* __restrict__ tw, typename Cfg::T scale) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  constexpr int R = Seq::r[P], Ns = Seq::ns(P);
  sfor<0, Cfg::bpt(P)>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    if constexpr (P != 0) hxw_twiddle<Cfg, P>(v[i], j % Ns, twl, tw);
    dft<R>(v[i]);
  });
  if constexpr (P == Cfg::NP - 1) {
    sfor<0, Cfg::bpt(P)>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      const unsigned j = tid + i * Cfg::TPF;
      const unsigned base = (j / Ns) * (Ns * R) + j % Ns;
      sfor<0, R>([&](auto u_) PFA_LAMBDA {
        constexpr int u = decltype(u_)::value;
        cx<T> y = v[i][u];
        if constexpr (BWD) y.im = -y.im;
        y.re *= scale;
        y.im *= scale;
        io.store(y, io.out_off(0, base), io.out_step(u * Ns));
      });
    });
  } else {
    cx<T> n[Cfg::bpt(P + 1)][Seq::r[P + 1]];
    hxw_exchange<Cfg, P>(v, n, tid, img);
    hxw_passes<Cfg, BWD, P + 1>(n, io, tid, img, twl, tw, scale);
  }
}

/// Body shared by the interleaved and the split-storage kernels (`make_io(g)`: the transform's I/O object)
template <typename Cfg, bool BWD, typename MakeIO>
PFA_DEV void stockham_wg_hx_body(MakeIO&& make_io, const cx<typename Cfg::T>* __restrict__ tw, long long nfft,
                                 typename Cfg::T scale) {
  using T = typename Cfg::T;
  static_assert(wg_hx_supported<Cfg>(), "see wg_hx_supported()");
  extern __shared__ __attribute__((aligned(16))) char pfa_smem[];
  cx<T>* img = reinterpret_cast<cx<T>*>(pfa_smem);
  cx<T>* twl = img + wg_hx_image_elems<Cfg>();
  const unsigned tid = threadIdx.x;
  if constexpr (Cfg::TWL > 0) {
    for (int i = threadIdx.x; i < Cfg::TWL_ELEMS; i += Cfg::WG) twl[i] = tw[i];
    __syncthreads();
  }
  for (long long g = blockIdx.x; g < nfft; g += gridDim.x) {
    const auto io = make_io(g);
    cx<T> v[Cfg::bpt(0)][Cfg::Seq::r[0]];
    wg_pass0_load<Cfg, BWD>(io, 0, static_cast<int>(tid), v);
    const cx<T>* twp = tw;
    asm volatile("" : "+s"(twp));  // keep the table reads inside the loop (see stockham_wg_body)
    hxw_passes<Cfg, BWD, 0>(v, io, tid, img, twl, twp, scale);
  }
}

template <typename Cfg, bool BWD>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_wg_hx_kernel(const cx<typename Cfg::T>* in,
                                                                           cx<typename Cfg::T>* out,
                                                                           const cx<typename Cfg::T>* __restrict__ tw,
                                                                           long long nfft, typename Cfg::T scale) {
  using T = typename Cfg::T;
  stockham_wg_hx_body<Cfg, BWD>(
      [&](long long g) PFA_LAMBDA { return packed_io<T, Cfg::N, 1, Cfg::AUX>(in, out, g, nfft); }, tw, nfft, scale);
}

template <typename Cfg, bool BWD>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_wg_hx_split_kernel(
    const typename Cfg::T* in_re, const typename Cfg::T* in_im, typename Cfg::T* out_re, typename Cfg::T* out_im,
    const cx<typename Cfg::T>* __restrict__ tw, long long nfft, typename Cfg::T scale) {
  using T = typename Cfg::T;
  stockham_wg_hx_body<Cfg, BWD>(
      [&](long long g) PFA_LAMBDA { return packed_split_io<T, Cfg::N, 1, Cfg::AUX>(in_re, in_im, out_re, out_im, g, nfft); },
      tw, nfft, scale);
}

}  // namespace pfa
