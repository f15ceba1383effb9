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
// through an image of (about) HALF the transform.  Between pass p (radix R, stride Ns) and pass p + 1 (radix R1, NB1 =
// N / R1 butterflies) the elements split at H = ceil(R1 / 2) * NB1: round 0 moves [0, H), round 1 moves [H, N).  On the
// reading side that is a split by butterfly leg (legs t < ceil(R1 / 2) read in round 0), on the writing side a split by
// butterfly (H is a multiple of the Ns * R elements a butterfly's outputs span, so butterfly j writes in round 0 iff
// j < H / R) -- any radices, ragged passes included; for even R1 the two halves are equal.  The registers a lane frees
// by writing are the ones it reads into, so the peak is the transform itself plus a butterfly's temporaries.  64-bit
// (fp64: 128-bit) LDS accesses at their full rate, four barriers per exchange instead of two.
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

/// Is the configuration one this kernel can run?  One transform per work-group, direct HBM access, two passes at least.
template <typename Cfg>
constexpr bool wg_hx_supported() {
  return Cfg::NP >= 2 && Cfg::FPW == 1 && Cfg::STAGED == 0;
}

/// the split point of the exchange behind pass P: elements [0, H) move in round 0, [H, N) in round 1
template <typename Cfg>
constexpr int wg_hx_split(int P) {
  const int r1 = Cfg::Seq::r[P + 1];
  return ((r1 + 1) / 2) * (Cfg::N / r1);
}

/// elements of the image (the largest first half of any exchange, padded like wg_cfg's full image), and the LDS bytes
template <typename Cfg>
constexpr int wg_hx_image_elems() {
  int h = 0;
  for (int p = 0; p + 1 < Cfg::NP; ++p) h = wg_hx_split<Cfg>(p) > h ? wg_hx_split<Cfg>(p) : h;
  return Cfg::pad(h - 1) + 1 + (Cfg::PADS == 0 ? 0 : Cfg::PADW);
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

/// exchange between pass P and pass P + 1 through the image, in two rounds
template <typename Cfg, int P>
PFA_DEV void hxw_exchange(cx<typename Cfg::T> (&v)[Cfg::bpt(P)][Cfg::Seq::r[P]],
                          cx<typename Cfg::T> (&n)[Cfg::bpt(P + 1)][Cfg::Seq::r[P + 1]], unsigned tid,
                          cx<typename Cfg::T>* img) {
  using Seq = typename Cfg::Seq;
  constexpr int R = Seq::r[P], Ns = Seq::ns(P), NB = Cfg::N / R, BPT = Cfg::bpt(P);
  constexpr int R1 = Seq::r[P + 1], NB1 = Cfg::N / R1, BPT1 = Cfg::bpt(P + 1);
  constexpr int HL = (R1 + 1) / 2;         // legs of a pass-(P + 1) butterfly read in round 0
  constexpr int H = wg_hx_split<Cfg>(P);   // = HL * NB1, a multiple of Ns * R
  constexpr int J0 = H / R;                // butterflies of pass P that write in round 0
  static_assert(H % (Ns * R) == 0 && J0 * R == H, "the split point lies between two butterflies' outputs");
  sfor<0, 2>([&](auto h_) PFA_LAMBDA {
    constexpr int h = decltype(h_)::value;
    sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      const unsigned j = tid + i * Cfg::TPF;
      // butterflies [lo, hi) write in this round; most (lane, i) slots are wholly inside or wholly outside at compile time
      constexpr int lo = h == 0 ? 0 : J0, hi = h == 0 ? J0 : NB;
      constexpr bool none = (i + 1) * Cfg::TPF <= lo || i * Cfg::TPF >= hi;
      constexpr bool all = i * Cfg::TPF >= lo && (i + 1) * Cfg::TPF <= hi;
      if constexpr (!none) {
        if (all || (j >= static_cast<unsigned>(lo) && j < static_cast<unsigned>(hi))) {
          const unsigned base = (j / Ns) * (Ns * R) + j % Ns - h * H;
          if constexpr (pad_is_linear<Cfg>(Ns, R, Ns * R) && (H % (Cfg::PADS == 0 ? 1 : Cfg::PADS) == 0)) {
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
      }
    });
    __syncthreads();
    sfor<0, BPT1>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      // (a slot that runs past the last butterfly reads butterfly NB1 - 1 again instead of being predicated: straight-line
      //  code -- a predicated read of 16 values into live registers made the allocator keep both versions; slots wholly
      //  past the end read nothing)
      constexpr bool none1 = i * Cfg::TPF >= NB1, all1 = (i + 1) * Cfg::TPF <= NB1;
      if constexpr (!none1) {
        unsigned j = tid + i * Cfg::TPF;
        if constexpr (!all1) j = j < static_cast<unsigned>(NB1) ? j : static_cast<unsigned>(NB1 - 1);
        constexpr int t0 = h == 0 ? 0 : HL, t1 = h == 0 ? HL : R1;
        if constexpr (pad_is_linear<Cfg>(NB1, R1, 1)) {
          const cx<typename Cfg::T>* p = img + lds_pad<Cfg>(j);
          sfor<t0, t1>([&](auto t_) PFA_LAMBDA {
            constexpr int t = decltype(t_)::value;
            n[i][t] = p[(t - t0) * pad_step<Cfg>(NB1)];
          });
        } else {
          sfor<t0, t1>([&](auto t_) PFA_LAMBDA {
            constexpr int t = decltype(t_)::value;
            n[i][t] = img[lds_pad<Cfg>(j + (t - t0) * NB1)];
          });
        }
      }
    });
    __syncthreads();
  });
}

/// `after_first_exchange(v)`: called once the exchange behind pass 0 has moved pass 0's outputs out of `v` -- the
/// software-pipelined kernel (PF = 1) issues the NEXT transform's loads into those registers there.
/// (Round 5's LDS-DMA form -- PF = 2: the next transform's first half on its way into the idle image behind the last
///  exchange, bit-identical, +0 ... 3 %, profiles/r5_hx_tune_lds_dma.txt -- left this header in round 6; its source is
///  stockham_wg_hx.hpp:234-288 of commit b38555f.)
template <typename Cfg, bool BWD, int P, typename IO, typename Hook>
PFA_DEV void hxw_passes(cx<typename Cfg::T> (&v)[Cfg::bpt(P)][Cfg::Seq::r[P]], const IO& io, unsigned tid,
                        cx<typename Cfg::T>* img, const cx<typename Cfg::T>* twl,
                        const cx<typename Cfg::T>* __restrict__ tw, typename Cfg::T scale, Hook&& after_first_exchange) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  constexpr int R = Seq::r[P], Ns = Seq::ns(P);
  sfor<0, Cfg::bpt(P)>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    // (no `j < NB` around the arithmetic of a ragged pass: the idle lanes compute on whatever their registers hold --
    //  their table addresses are valid, nothing of theirs is written -- and the butterflies stay straight-line code; a
    //  predicated in-place update kept both versions of the lane's 30-60 values alive: 30.10.10.10 on 1024 lanes spilled
    //  176 registers, on 1000 lanes none)
    if constexpr (P != 0) hxw_twiddle<Cfg, P>(v[i], j % Ns, twl, tw);
    dft<R>(v[i]);
  });
  if constexpr (P == Cfg::NP - 1) {
    sfor<0, Cfg::bpt(P)>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      const unsigned j = tid + i * Cfg::TPF;
      const unsigned base = (j / Ns) * (Ns * R) + j % Ns;
      constexpr bool none_s = i * Cfg::TPF >= Cfg::N / R, all_s = (i + 1) * Cfg::TPF <= Cfg::N / R;
      if constexpr (!none_s) {
        // (the lanes of a slot that runs past the last butterfly store out of the buffer's range: dropped by the hardware)
        const unsigned voff = (all_s || j < static_cast<unsigned>(Cfg::N / R)) ? io.out_off(0, base) : 0xFFFFFFF0u;
        sfor<0, R>([&](auto u_) PFA_LAMBDA {
          constexpr int u = decltype(u_)::value;
          cx<T> y = v[i][u];
          if constexpr (BWD) y.im = -y.im;
          y.re *= scale;
          y.im *= scale;
          io.store(y, voff, io.out_step(u * Ns));
        });
      }
    });
  } else {
    cx<T> n[Cfg::bpt(P + 1)][Seq::r[P + 1]];
    hxw_exchange<Cfg, P>(v, n, tid, img);
    if constexpr (P == 0) after_first_exchange(v);
    hxw_passes<Cfg, BWD, P + 1>(n, io, tid, img, twl, tw, scale, after_first_exchange);
  }
}

/// pass 0's loads (legs T0 and up); the lanes of a slot past the last butterfly read out of the buffer's range (zeros),
/// unpredicated
template <typename Cfg, bool BWD, int T0 = 0, typename IO>
PFA_DEV void hxw_load(const IO& io, unsigned tid, cx<typename Cfg::T> (&v)[Cfg::bpt(0)][Cfg::Seq::r[0]]) {
  using T = typename Cfg::T;
  constexpr int R = Cfg::Seq::r[0], NB = Cfg::N / R;
  sfor<0, Cfg::bpt(0)>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    constexpr bool none = i * Cfg::TPF >= NB, all = (i + 1) * Cfg::TPF <= NB;
    if constexpr (!none) {
      const unsigned voff = (all || j < static_cast<unsigned>(NB)) ? io.in_off(0, j) : 0xFFFFFFF0u;
      sfor<T0, R>([&](auto t_) PFA_LAMBDA {
        constexpr int t = decltype(t_)::value;
        cx<T> x = io.load(voff, io.in_step(t * NB));
        if constexpr (BWD) x.im = -x.im;
        v[i][t] = x;
      });
    }
  });
}

/// Body shared by the interleaved and the split-storage kernels (`make_io(g)`: the transform's I/O object).
/// PF: software-pipelined -- the next transform's HBM loads are issued behind the first exchange, into the registers
/// pass 0 has just vacated, and are in flight through the remaining passes (for transforms that leave a lane that many
/// registers: 2 x the transform + a butterfly's temporaries; fp64 8192 on 512 lanes, fp32 16384 on 1024)
template <typename Cfg, bool BWD, int PF = 0, typename MakeIO>
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
  cx<T> v[Cfg::bpt(0)][Cfg::Seq::r[0]];
  if constexpr (PF == 1) {
    if (static_cast<long long>(blockIdx.x) < nfft) hxw_load<Cfg, BWD>(make_io(blockIdx.x), tid, v);
  }
  for (long long g = blockIdx.x; g < nfft; g += gridDim.x) {
    const auto io = make_io(g);
    if constexpr (PF == 0) hxw_load<Cfg, BWD>(io, tid, v);
    const cx<T>* twp = tw;
    asm volatile("" : "+s"(twp));  // keep the table reads inside the loop (see stockham_wg_body)
    const long long gn = g + gridDim.x;
    hxw_passes<Cfg, BWD, 0>(v, io, tid, img, twl, twp, scale,
                            [&](cx<T> (&regs)[Cfg::bpt(0)][Cfg::Seq::r[0]]) PFA_LAMBDA {
                              if constexpr (PF == 1) {
                                if (gn < nfft) hxw_load<Cfg, BWD>(make_io(gn), tid, regs);
                              }
                            });
  }
}

template <typename Cfg, bool BWD, int PF = 0>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_wg_hx_kernel(const cx<typename Cfg::T>* in,
                                                                           cx<typename Cfg::T>* out,
                                                                           const cx<typename Cfg::T>* __restrict__ tw,
                                                                           long long nfft, typename Cfg::T scale) {
  using T = typename Cfg::T;
  stockham_wg_hx_body<Cfg, BWD, PF>(
      [&](long long g) PFA_LAMBDA { return packed_io<T, Cfg::N, 1, Cfg::AUX>(in, out, g, nfft); }, tw, nfft, scale);
}

template <typename Cfg, bool BWD, int PF = 0>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_wg_hx_split_kernel(
    const typename Cfg::T* in_re, const typename Cfg::T* in_im, typename Cfg::T* out_re, typename Cfg::T* out_im,
    const cx<typename Cfg::T>* __restrict__ tw, long long nfft, typename Cfg::T scale) {
  using T = typename Cfg::T;
  stockham_wg_hx_body<Cfg, BWD, PF>(
      [&](long long g) PFA_LAMBDA { return packed_split_io<T, Cfg::N, 1, Cfg::AUX>(in_re, in_im, out_re, out_im, g, nfft); },
      tw, nfft, scale);
}

}  // namespace pfa
