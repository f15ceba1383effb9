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

/// Storage of the two sides of a strided stage: SPLIT is 0 (both interleaved), 1 (both split: real + imaginary
/// plane), 2 (split input, interleaved output) or 3 (interleaved input, split output) -- the mixed forms connect
/// SPLIT_COMPLEX user buffers to the interleaved scratch of the four-step tier.
constexpr bool split_in(int mode) { return mode == 1 || mode == 2; }
constexpr bool split_out(int mode) { return mode == 1 || mode == 3; }

/// The descriptors of one group: interleaved (one per side) or split (real + imaginary plane per side).
template <typename T, int AUX, int SPLIT>
struct strided_io {
  static constexpr unsigned ES_IN = split_in(SPLIT) ? sizeof(T) : sizeof(cx<T>);
  static constexpr unsigned ES_OUT = split_out(SPLIT) ? sizeof(T) : sizeof(cx<T>);
  using off_t = unsigned;  // type of a butterfly leg's uniform byte offset (strided_io_big: 64 bits)
  __amdgpu_buffer_rsrc_t rin, rout, rin_im, rout_im;
  PFA_DEV cx<T> load(unsigned voff, unsigned soff) const {
    if constexpr (split_in(SPLIT)) {
      return {buf_load_scalar<T, AUX>(rin, voff, soff), buf_load_scalar<T, AUX>(rin_im, voff, soff)};
    } else {
      return buf_load<T, AUX>(rin, voff, soff);
    }
  }
  PFA_DEV void store(cx<T> v, unsigned voff, unsigned soff) const {
    if constexpr (split_out(SPLIT)) {
      buf_store_scalar<T, AUX>(v.re, rout, voff, soff);
      buf_store_scalar<T, AUX>(v.im, rout_im, voff, soff);
    } else {
      buf_store<T, AUX>(v, rout, voff, soff);
    }
  }
};

/// (no standard library under hiprtc: std::conditional, spelled out)
template <bool C, typename A, typename B>
struct pick_type {
  using type = A;
};
template <typename A, typename B>
struct pick_type<false, A, B> {
  using type = B;
};

/// BIG groups (round 6): a group whose elements span 4 GiB or more -- a batch-interleaved array of that size: element i of
/// a transform sits i * batch elements from element 0 -- cannot be addressed through one buffer descriptor: the uniform
/// offset of a butterfly leg (the scalar offset of the buffer instructions) no longer fits 32 bits.  This I/O object has the
/// interface of strided_io with 64-bit leg offsets and plain global accesses: lane offset (32 bits: the span of the first
/// pass's butterflies, 1 / R0 of the group) + leg offset (64 bits) + the group's base pointer.  Dead lanes (offset
/// 0xFFFFFFF0, see strided_pass) load zeros and store nothing.  Only the kernels instantiated with BIG use it (compiled at
/// commit for such arrays: plan_batch_interleaved_two_stage); every other kernel's code is unchanged.
template <typename T, int AUX, int SPLIT>
struct strided_io_big {
  static constexpr unsigned ES_IN = split_in(SPLIT) ? sizeof(T) : sizeof(cx<T>);
  static constexpr unsigned ES_OUT = split_out(SPLIT) ? sizeof(T) : sizeof(cx<T>);
  using off_t = unsigned long long;
  using vec_t = typename pick_type<sizeof(T) == 4, buf_b64_t, buf_b128_t>::type;
  const char* in;
  const char* in_im;
  char* out;
  char* out_im;
  // (AUX == 0: the default cache policy on both sides -- policy 3 of aux_of_policy, unaligned row pitches; else streamed)
  template <typename V>
  static PFA_DEV V ld(const V* p) {
    if constexpr (AUX == 0) {
      return *p;
    } else {
      return __builtin_nontemporal_load(p);
    }
  }
  template <typename V>
  static PFA_DEV void st(V v, V* p) {
    if constexpr (AUX == 0) {
      *p = v;
    } else {
      __builtin_nontemporal_store(v, p);
    }
  }
  PFA_DEV cx<T> load(unsigned voff, off_t soff) const {
    cx<T> x = {T(0), T(0)};
    if (voff != 0xFFFFFFF0u) {
      if constexpr (split_in(SPLIT)) {
        x.re = ld(reinterpret_cast<const T*>(in + soff + voff));
        x.im = ld(reinterpret_cast<const T*>(in_im + soff + voff));
      } else {
        x = __builtin_bit_cast(cx<T>, ld(reinterpret_cast<const vec_t*>(in + soff + voff)));
      }
    }
    return x;
  }
  PFA_DEV void store(cx<T> v, unsigned voff, off_t soff) const {
    if (voff != 0xFFFFFFF0u) {
      if constexpr (split_out(SPLIT)) {
        st(v.re, reinterpret_cast<T*>(out + soff + voff));
        st(v.im, reinterpret_cast<T*>(out_im + soff + voff));
      } else {
        st(__builtin_bit_cast(vec_t, v), reinterpret_cast<vec_t*>(out + soff + voff));
      }
    }
  }
};

/// ROW_IN / ROW_OUT: that side of the group is row shaped (each FFT contiguous) and is copied HBM <-> LDS with
/// element-fastest lanes (full lines) instead of being addressed f-fastest by the passes; the LDS image then uses an
/// odd pitch (FPW + 1) so that both the element-fastest copy and the f-fastest passes are bank-conflict free.
template <typename Cfg, bool ROW>
constexpr int strided_pitch() {
  return ROW ? Cfg::FPW + 1 : Cfg::FPW;
}

/// LDS copy of the store-modifier tables (strided_args::stw_tab): behind the images and the TWL tables.
/// (IMGDIV: the kernel's image holds 1 / IMGDIV of the group -- 2 for the half-exchange kernel of
/// tools/probes/stockham_strided_hx.hpp, the first radix for the experiment of tools/probes/stockham_strided_sfr.hpp)
template <typename Cfg, int IMGDIV = 1>
PFA_DEV cx<typename Cfg::T>* stw_lds_tables(const strided_args& a) {
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  if (a.stw_lds_off != 0) return reinterpret_cast<cx<typename Cfg::T>*>(pfa_smem_strided + a.stw_lds_off);
  constexpr size_t image = size_t(Cfg::N) * Cfg::FPW / IMGDIV;
  constexpr size_t own = Cfg::NP > 1 ? image + Cfg::TWL_ELEMS : 0;
  return reinterpret_cast<cx<typename Cfg::T>*>(pfa_smem_strided) + own;
}

/// W_M^m as the product of one entry per level
template <typename Cfg, int IMGDIV = 1>
PFA_DEV cx<typename Cfg::T> stw_from_lds(const strided_args& a, unsigned m) {
  const cx<typename Cfg::T>* tab = stw_lds_tables<Cfg, IMGDIV>(a);
  const unsigned sh = static_cast<unsigned>(a.stw_lshift);
  const unsigned mask = (1u << sh) - 1u;
  cx<typename Cfg::T> w = tab[m & mask];
  if (a.stw_levels > 1) w = cmul(w, tab[(1u << sh) + ((m >> sh) & mask)]);
  if (a.stw_levels > 2) w = cmul(w, tab[(2u << sh) + ((m >> (2 * sh)) & mask)]);
  if (a.stw_levels > 3) w = cmul(w, tab[(3u << sh) + ((m >> (3 * sh)) & mask)]);
  return w;
}

/// once per work-group lifetime (STW kernels): the tables into LDS
template <typename Cfg, int STW, int IMGDIV = 1>
PFA_DEV void strided_copy_stw(const strided_args& a) {
  if constexpr (STW == 1) {
    cx<typename Cfg::T>* dst = stw_lds_tables<Cfg, IMGDIV>(a);
    const cx<typename Cfg::T>* src = static_cast<const cx<typename Cfg::T>*>(a.stw_tab);
    const int n = a.stw_levels << a.stw_lshift;
    for (int i = threadIdx.x; i < n; i += Cfg::WG) dst[i] = src[i];
    __syncthreads();
  }
}

/// v[u] *= W_M^{(base + u * stride) * c}, u < R -- the inter-stage twiddles of the four-step decomposition, for the
/// butterfly outputs of a stage A's last pass (base = element index, stride = Ns) or for the butterfly inputs of a
/// stage B's first pass (base = j, stride = N / R: strided_pass0_compute LTW).  W^{base*c}, the step W^{stride*c} and
/// every fourth power of the step come from the tables, the other powers are one multiply away from those.
/// Exponents in 32 bits: k * c < n1 * n2 = M <= 2^28 for every element k of a stage and every column c.
template <typename Cfg, int STW, int R, int IMGDIV = 1>
PFA_DEV void stw_apply(const strided_args& a, unsigned base, unsigned stride, unsigned stw_c, cx<typename Cfg::T> (&v)[R]) {
  using T = typename Cfg::T;
  static_assert(STW != 0, "no modifier requested");
  const unsigned m0 = base * stw_c;
  const unsigned ms = stride * stw_c;
  // W^m: STW == 1 from the small multi-level tables in LDS (strided_copy_stw) -- two L2-resident tables read with
  // scattered 16-byte gathers cost the pre-compiled stage kernels 4.5 % (C3 stage A) to 15 % (fp32 n = 1024) --,
  // STW == 2 from those two global tables (hi/lo split of the exponent): kernels that are bound by their
  // arithmetic and LDS traffic, not by the vector memory path (runtime-planned odd radices with several
  // work-groups per CU: fp32 N = 30000 stage A 144 us with gathers, 185 us with three-level LDS tables)
  auto root = [&](unsigned m) PFA_LAMBDA -> cx<T> {
    if constexpr (STW == 2) {
      const cx<T>* lo = static_cast<const cx<T>*>(a.stw_lo);
      const cx<T>* hi = static_cast<const cx<T>*>(a.stw_hi);
      return cmul(lo[m & ((1u << a.stw_shift) - 1u)], hi[m >> a.stw_shift]);
    } else {
      return stw_from_lds<Cfg, IMGDIV>(a, m);
    }
  };
  const cx<T> w0 = root(m0);
  // stw[u] = W^{(base + u*stride)*c} = w0 * step^u.  Every fourth one is w0 times a table value (anchor), the
  // three behind it are one multiply away from their anchor -- a chain of squarings of the step would amplify
  // its rounding by u (radix 32: ~45 ulp in fp32), and folding w0 into the anchors saves R multiplies.
  cx<T> stw[R];
  [[maybe_unused]] cx<T> pw[4];
  if constexpr (R > 1) pw[1] = root(ms);
  if constexpr (R > 2) pw[2] = cmul(pw[1], pw[1]);
  if constexpr (R > 3) pw[3] = cmul(pw[2], pw[1]);
  sfor<0, (R + 3) / 4>([&](auto k_) PFA_LAMBDA {
    constexpr int k = decltype(k_)::value;
    cx<T> anchor = w0;
    if constexpr (k > 0) {
      const unsigned mu = ms * static_cast<unsigned>(4 * k);
      anchor = cmul(w0, root(mu));
    }
    stw[4 * k] = anchor;
    sfor<1, 4>([&](auto r_) PFA_LAMBDA {
      constexpr int r = decltype(r_)::value;
      if constexpr (4 * k + r < R) stw[4 * k + r] = cmul(anchor, pw[r]);
    });
  });
  sfor<0, R>([&](auto u_) PFA_LAMBDA {
    constexpr int u = decltype(u_)::value;
    v[u] = cmul(v[u], stw[u]);
  });
}

/// column index of the store / load modifier of FFT f of a group (strided_args::stw_cdiv: 1 for packed data)
PFA_DEV unsigned stw_column(const strided_args& a, long long c0, unsigned f) {
  if (a.stw_cdiv > 1) {
    return static_cast<unsigned>(static_cast<unsigned long long>(c0 + f) / static_cast<unsigned long long>(a.stw_cdiv));
  }
  return static_cast<unsigned>(c0 + f);
}

/// The HBM side of a last pass: butterfly outputs v[u] = element (base + u * Ns) of FFT f go to memory, conjugated
/// for the backward transform, scaled, and -- STW -- multiplied by the store modifier W_M^{k*c}.
template <typename Cfg, bool BWD, int STW, int R, int Ns, typename IO, int IMGDIV = 1>
PFA_DEV void strided_store_butterfly(const IO& io, const strided_args& a, unsigned f, unsigned base, bool live,
                                     long long c0, cx<typename Cfg::T> (&v)[R]) {
  using T = typename Cfg::T;
  constexpr unsigned ES_OUT = IO::ES_OUT;
  const unsigned osh = static_cast<unsigned>(a.out_tile_shift);
  const unsigned omul = a.out_tile_mul != 0 ? a.out_tile_mul : 1u;
  const unsigned voff =
      live ? (f * a.out_fdist + (base >> osh) * a.out_stride + (base & ((1u << osh) - 1u)) * omul) * ES_OUT
           : 0xFFFFFFF0u;
  const T scale = static_cast<T>(a.scale);
  if constexpr (STW != 0) stw_apply<Cfg, STW, R, IMGDIV>(a, base, static_cast<unsigned>(Ns), stw_column(a, c0, f), v);
  sfor<0, R>([&](auto u_) PFA_LAMBDA {
    constexpr int u = decltype(u_)::value;
    cx<T> y = v[u];
    if constexpr (BWD) y.im = -y.im;
    y.re *= scale;
    y.im *= scale;
    io.store(y, voff, static_cast<typename IO::off_t>(static_cast<unsigned>(u * Ns) >> osh) * a.out_stride * ES_OUT);
  });
}

/// TIN (tiled input, the four-step stage B behind a group-major stage A): the input of the group is a sequence of
/// tiles [f][i % TW] of FPW x TW elements (strided_args::in_tile_shift == log2 TW; TW = the group width of the stage A
/// that wrote them).  Addressed f-fastest a wave would read one whole tile per instruction but with its lanes transposed
/// inside it (16-byte pieces 128 B apart: 5.1 instead of 5.8 TB/s on the C3 stage B, profiles/r2_notes.md).  With TIN
/// pass 0 takes its lanes element-fastest inside a tile -- lane = (i % TW) + TW * f + TW * FPW * (i / TW) -- so every
/// wave-instruction reads consecutive addresses, and the exchange behind pass 0 stores f ^ (i % FPW) in place of f so
/// that the scatter (lane stride R0 elements = a multiple of all banks) stays conflict-free (TW <= FPW; two-way
/// conflicts when TW = 2 * FPW); pass 1 reads through the same permutation, the later passes are unchanged.
/// Template value: 0 off, 1 square tiles (TW = FPW), any other positive value = TW (a stage A with wider groups: fp32
/// n2 = 2048 holds 8 columns, its stage A 16).  Needs TW * FPW | WG, (N / R0) % TW == 0 and (N / R0) % TPF == 0.
/// (Round 5's ROWS form -- TIN = -1: lanes element-fastest over a whole row of a row-major intermediate, +5 % for N = 10^6,
///  -3 ... +2 % on 20 other lengths, profiles/r5_perf_tin_rows.txt -- left this header in round 6; its source is
///  stockham_strided.hpp of commit b38555f.)
template <typename Cfg, int TIN = 1>
constexpr int tin_width() {
  return TIN == 1 ? Cfg::FPW : TIN;
}
template <typename Cfg, int TIN = 1>
constexpr bool tin_supported() {
  constexpr int TW = tin_width<Cfg, TIN>();
  return TIN != 0 && Cfg::NP >= 2 && (Cfg::FPW & (Cfg::FPW - 1)) == 0 && (TW & (TW - 1)) == 0 && TW >= 2 &&
         Cfg::WG % (TW * Cfg::FPW) == 0 && (Cfg::N / Cfg::Seq::r[0]) % TW == 0 &&
         (Cfg::N / Cfg::Seq::r[0]) % Cfg::TPF == 0;
}

template <typename Cfg, bool BWD, int STW, int P, typename IO, bool ROW_IN = false, bool ROW_OUT = false,
          int TIN = 0>
PFA_DEV void strided_pass(const IO& io, const strided_args& a, unsigned f,
                          unsigned tid, bool live, long long c0, cx<typename Cfg::T>* lds,
                          const cx<typename Cfg::T>* __restrict__ tw, long long nlive = 0) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  constexpr int R = Seq::r[P];
  constexpr int N = Cfg::N;
  constexpr int NB = N / R;
  constexpr int Ns = Seq::ns(P);
  constexpr int BPT = Cfg::bpt(P);
  constexpr bool ragged = (NB % Cfg::TPF) != 0;
  constexpr bool first = P == 0 && !ROW_IN;   // reads HBM directly
  constexpr bool last = P == Cfg::NP - 1 && !ROW_OUT;  // writes HBM directly
  constexpr int FPW = strided_pitch<Cfg, ROW_IN || ROW_OUT>();
  [[maybe_unused]] constexpr unsigned ES_IN = IO::ES_IN;

  constexpr bool tin0 = TIN != 0 && P == 0;  // lanes element-fastest inside the input tiles
  constexpr bool tin1 = TIN != 0 && P == 1;  // reads through the permutation pass 0 stored with
  [[maybe_unused]] unsigned tin_jl = 0;
  if constexpr (tin0) {
    static_assert(first && tin_supported<Cfg, TIN>(), "TIN: see tin_supported()");
    const unsigned lane = threadIdx.x;
    constexpr unsigned TW = tin_width<Cfg, TIN>();
    tin_jl = lane % TW;
    f = (lane / TW) % Cfg::FPW;
    tid = (lane / (TW * Cfg::FPW)) * TW + tin_jl;
    live = static_cast<long long>(f) < nlive;
  }
  // LDS copy of the leading twiddle tables: behind the image, unless the launch says otherwise (strided_args::twl_lds_off)
  [[maybe_unused]] const cx<T>* twl = lds + N * FPW;
  if constexpr (P != 0 && P <= Cfg::TWL && !ROW_IN && !ROW_OUT) {
    extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
    if (a.twl_lds_off != 0) twl = reinterpret_cast<const cx<T>*>(pfa_smem_strided + a.twl_lds_off);
  }
  cx<T> v[BPT][R];
  sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    if (!ragged || j < NB) {
      if constexpr (first) {
        // dead lanes (FFT beyond the end) get an out-of-range offset: the buffer range check returns zeros
        const unsigned tsh = static_cast<unsigned>(a.in_tile_shift);
        const unsigned voff =
            live ? (f * a.in_fdist + (j >> tsh) * a.in_stride + (j & ((1u << tsh) - 1u))) * ES_IN : 0xFFFFFFF0u;
        sfor<0, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          cx<T> x = io.load(voff, static_cast<typename IO::off_t>(static_cast<unsigned>(t * NB) >> tsh) * a.in_stride * ES_IN);
          if constexpr (BWD) x.im = -x.im;
          v[i][t] = x;
        });
      } else if constexpr (tin1) {
        constexpr int R0 = Seq::r[0];
        if constexpr (NB % (R0 * Cfg::FPW) == 0) {  // the permutation does not depend on the butterfly leg
          const cx<T>* p = lds + j * FPW + (f ^ ((j / R0) % Cfg::FPW));
          sfor<0, R>([&](auto t_) PFA_LAMBDA {
            constexpr int t = decltype(t_)::value;
            v[i][t] = p[t * NB * FPW];
          });
        } else {
          sfor<0, R>([&](auto t_) PFA_LAMBDA {
            constexpr int t = decltype(t_)::value;
            const unsigned e = j + t * NB;
            v[i][t] = lds[e * FPW + (f ^ ((e / R0) % Cfg::FPW))];
          });
        }
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
      if constexpr (P != 0) {
        sfor<1, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          cx<T> w;
          if constexpr (P <= Cfg::TWL && !ROW_IN && !ROW_OUT) {
            w = (twl + Seq::tw_off(P) + (t - 1) * Ns)[q];  // LDS copy behind the image (wg_cfg TWL)
          } else {
            w = (tw + Seq::tw_off(P) + (t - 1) * Ns)[q];
          }
          v[i][t] = cmul(v[i][t], w);
        });
      }
      dft<R>(v[i]);
      const unsigned base = (j / Ns) * (Ns * R) + q;
      if constexpr (last) {
        strided_store_butterfly<Cfg, BWD, STW, R, Ns>(io, a, f, base, live, c0, v[i]);
      } else {
        // TIN pass 0: element e = j * R0 + u of FFT f goes to slot f ^ ((e / R0) % FPW) = f ^ (j % FPW)
        cx<T>* p = lds + base * FPW + (tin0 ? (f ^ (j % Cfg::FPW)) : f);
        sfor<0, R>([&](auto u_) PFA_LAMBDA {
          constexpr int u = decltype(u_)::value;
          p[u * Ns * FPW] = v[i][u];
        });
      }
    }
  });
  if constexpr (!last) __syncthreads();
}

template <typename Cfg, bool BWD, int STW, int P, typename IO, bool ROW_IN = false, bool ROW_OUT = false,
          int TIN = 0>
PFA_DEV void strided_passes(const IO& io, const strided_args& a, unsigned f, unsigned tid, bool live, long long c0,
                            cx<typename Cfg::T>* lds, const cx<typename Cfg::T>* __restrict__ tw,
                            long long nlive = 0) {
  if constexpr (P < Cfg::NP) {
    strided_pass<Cfg, BWD, STW, P, IO, ROW_IN, ROW_OUT, TIN>(io, a, f, tid, live, c0, lds, tw, nlive);
    strided_passes<Cfg, BWD, STW, P + 1, IO, ROW_IN, ROW_OUT, TIN>(io, a, f, tid, live, c0, lds, tw, nlive);
  }
}

/// passes [P, PEND) (the XCD-local four-step kernel puts its hand-off waits between passes)
template <typename Cfg, bool BWD, int STW, int P, int PEND, typename IO, int TIN = 0>
PFA_DEV void strided_passes_range(const IO& io, const strided_args& a, unsigned f, unsigned tid, bool live, long long c0,
                                  cx<typename Cfg::T>* lds, const cx<typename Cfg::T>* __restrict__ tw,
                                  long long nlive = 0) {
  if constexpr (P < PEND) {
    strided_pass<Cfg, BWD, STW, P, IO, false, false, TIN>(io, a, f, tid, live, c0, lds, tw, nlive);
    strided_passes_range<Cfg, BWD, STW, P + 1, PEND, IO, TIN>(io, a, f, tid, live, c0, lds, tw, nlive);
  }
}

/// TIN lane mapping of pass 0 (see strided_pass): lanes element-fastest inside the FPW x TW input tiles
template <typename Cfg, int TIN = 1>
PFA_DEV void tin_lanes(unsigned* f, unsigned* tid, bool* live, long long nlive) {
  const unsigned lane = threadIdx.x;
  constexpr unsigned TW = tin_width<Cfg, TIN>();
  *f = (lane / TW) % Cfg::FPW;
  *tid = (lane / (TW * Cfg::FPW)) * TW + lane % TW;
  *live = static_cast<long long>(*f) < nlive;
}

/// Pass 0 of the strided kernel split in two (loads / butterfly + scatter) for the prefetching variant.
/// (f, tid, live) are the TIN lane mapping (tin_lanes) when the kernel reads tiled input.
template <typename Cfg, bool BWD, typename IO>
PFA_DEV void strided_pass0_load(const IO& io, const strided_args& a, unsigned f, unsigned tid, bool live,
                                cx<typename Cfg::T> (&v)[Cfg::bpt(0)][Cfg::Seq::r[0]]) {
  using T = typename Cfg::T;
  constexpr int R = Cfg::Seq::r[0];
  constexpr int NB = Cfg::N / R;
  constexpr bool ragged = (NB % Cfg::TPF) != 0;
  constexpr unsigned ES = IO::ES_IN;
  sfor<0, Cfg::bpt(0)>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    if (!ragged || j < NB) {
      const unsigned tsh = static_cast<unsigned>(a.in_tile_shift);
      const unsigned voff =
          live ? (f * a.in_fdist + (j >> tsh) * a.in_stride + (j & ((1u << tsh) - 1u))) * ES : 0xFFFFFFF0u;
      sfor<0, R>([&](auto t_) PFA_LAMBDA {
        constexpr int t = decltype(t_)::value;
        cx<T> x = io.load(voff, static_cast<typename IO::off_t>(static_cast<unsigned>(t * NB) >> tsh) * a.in_stride * ES);
        if constexpr (BWD) x.im = -x.im;
        v[i][t] = x;
      });
    }
  });
}

/// LTW (1 LDS tables / 2 global tables, as STW): the inter-stage twiddles applied to the INPUTS of the stage -- element
/// j + t * NB of the FFT with inner index c0 + f is multiplied by W_M^{(j + t*NB) * c} before pass 0 (the four-step
/// stage B carrying the modifier instead of stage A's stores)
template <typename Cfg, int TIN = 0, int LTW = 0>
PFA_DEV void strided_pass0_compute(cx<typename Cfg::T> (&v)[Cfg::bpt(0)][Cfg::Seq::r[0]], unsigned f, unsigned tid,
                                   cx<typename Cfg::T>* lds, const strided_args* a = nullptr, long long c0 = 0) {
  constexpr int R = Cfg::Seq::r[0];
  constexpr int NB = Cfg::N / R;
  constexpr bool ragged = (NB % Cfg::TPF) != 0;
  sfor<0, Cfg::bpt(0)>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    if (!ragged || j < NB) {
      if constexpr (LTW != 0) stw_apply<Cfg, LTW, R>(*a, j, static_cast<unsigned>(NB), stw_column(*a, c0, f), v[i]);
      dft<R>(v[i]);
      // TIN: element e = j * R + u of FFT f goes to slot f ^ (j % FPW) (strided_pass)
      cx<typename Cfg::T>* p = lds + (j * R) * Cfg::FPW + (TIN != 0 ? (f ^ (j % Cfg::FPW)) : f);
      sfor<0, R>([&](auto u_) PFA_LAMBDA {
        constexpr int u = decltype(u_)::value;
        p[u * Cfg::FPW] = v[i][u];
      });
    }
  });
  __syncthreads();
}

/// Groups never straddle an outer index: every outer index owns ceil(inner / FPW) groups, the last one possibly
/// partial (its surplus lanes are masked through `live`), so `inner` need not be a multiple of FPW.
template <typename Cfg>
PFA_DEV long long strided_ngroups(const strided_args& a) {
  const long long per_outer = (a.inner + Cfg::FPW - 1) / Cfg::FPW;
  return ((a.total + a.inner - 1) / a.inner) * per_outer;
}

template <typename Cfg, int SPLIT, bool BIG = false>
PFA_DEV auto strided_group(const strided_args& a, long long g, unsigned f, bool* live, long long* c0_out,
                           long long* nlive_out = nullptr, long long in_base = 0, long long out_base = 0) {
  using T = typename Cfg::T;
  using IO = typename pick_type<BIG, strided_io_big<T, Cfg::AUX, SPLIT>, strided_io<T, Cfg::AUX, SPLIT>>::type;
  constexpr unsigned ES_IN = IO::ES_IN, ES_OUT = IO::ES_OUT;
  const long long per_outer = (a.inner + Cfg::FPW - 1) / Cfg::FPW;
  const long long o = g / per_outer;
  const long long c0 = (g - o * per_outer) * Cfg::FPW;
  const long long t0 = o * a.inner + c0;
  const long long nlive = (a.inner - c0 < a.total - t0) ? a.inner - c0 : a.total - t0;
  *live = static_cast<long long>(f) < nlive;
#ifdef PFA_TUNE_DEAD_LANES  // tuners only: every lane masked -- no HBM traffic, the kernel's arithmetic / LDS time alone
  *live = false;
#endif
  *c0_out = c0;
  if (nlive_out != nullptr) *nlive_out = nlive;
  // group-major sides: group number (g - o * per_outer) of outer index o starts at that multiple of gdist
  const long long gw = g - o * per_outer;
  long long obase_in = o * a.in_dist_outer, obase_out = o * a.out_dist_outer;
  if (a.outer_lo > 0) {
    const long long ohi = o / a.outer_lo, olo = o - ohi * a.outer_lo;
    obase_in = ohi * a.in_dist_outer_hi + olo * a.in_dist_outer;
    obase_out = ohi * a.out_dist_outer_hi + olo * a.out_dist_outer;
  }
  // (in_base / out_base: the XCD-local four-step kernel places the scratch side of a group in a slot of its own choice)
  const long long ioff = in_base + obase_in + (a.in_gdist != 0 ? gw * a.in_gdist : c0 * a.in_fdist);
  const long long ooff = out_base + obase_out + (a.out_gdist != 0 ? gw * a.out_gdist : c0 * a.out_fdist);
  // ranges: last element of the last FFT of the group (the planner guarantees < 4 GiB)
  const unsigned in_bytes = (static_cast<unsigned>(Cfg::FPW - 1) * a.in_fdist +
                             (static_cast<unsigned>(Cfg::N - 1) >> a.in_tile_shift) * a.in_stride +
                             (1u << a.in_tile_shift)) * ES_IN;
  const unsigned out_bytes = (static_cast<unsigned>(Cfg::FPW - 1) * a.out_fdist +
                              (static_cast<unsigned>(Cfg::N - 1) >> a.out_tile_shift) * a.out_stride +
                              (1u << a.out_tile_shift) * (a.out_tile_mul != 0 ? a.out_tile_mul : 1u)) * ES_OUT;
  IO io;
  char* ip = const_cast<char*>(static_cast<const char*>(a.in)) + ioff * ES_IN;
  char* op = static_cast<char*>(a.out) + ooff * ES_OUT;
  if constexpr (BIG) {
    (void)in_bytes;
    (void)out_bytes;
    io.in = ip;
    io.out = op;
    io.in_im = split_in(SPLIT) ? static_cast<const char*>(a.in_im) + ioff * ES_IN : ip;
    io.out_im = split_out(SPLIT) ? static_cast<char*>(a.out_im) + ooff * ES_OUT : op;
    return io;
  } else {
  io.rin = __builtin_amdgcn_make_buffer_rsrc(ip, 0, in_bytes, 0x00020000);
  io.rout = __builtin_amdgcn_make_buffer_rsrc(op, 0, out_bytes, 0x00020000);
  if constexpr (split_in(SPLIT)) {
    char* ipi = const_cast<char*>(static_cast<const char*>(a.in_im)) + ioff * ES_IN;
    io.rin_im = __builtin_amdgcn_make_buffer_rsrc(ipi, 0, in_bytes, 0x00020000);
  } else {
    io.rin_im = io.rin;
  }
  if constexpr (split_out(SPLIT)) {
    char* opi = static_cast<char*>(a.out_im) + ooff * ES_OUT;
    io.rout_im = __builtin_amdgcn_make_buffer_rsrc(opi, 0, out_bytes, 0x00020000);
  } else {
    io.rout_im = io.rout;
  }
  return io;
  }
}

template <typename Cfg>
PFA_DEV void strided_copy_twiddles(cx<typename Cfg::T>* lds, const cx<typename Cfg::T>* __restrict__ tw);

/// Software-pipelined strided kernel: the loads of the work-group's next group are in flight during the LDS passes
/// of the current one (see stockham_wg_prefetch_kernel).  TIN: tiled input read with the lanes element-fastest inside
/// the tiles (strided_pass), for the four-step stage B behind a group-major stage A.
template <typename Cfg, bool BWD, int STW, int SPLIT = 0, int TIN = 0, int LTW = 0>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_strided_prefetch_kernel(const strided_args a) {
  using T = typename Cfg::T;
  static_assert(Cfg::NP >= 2, "the strided tier needs at least two passes (LDS exchange)");
  static_assert(TIN == 0 || tin_supported<Cfg, TIN>(), "TIN: see tin_supported()");
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem_strided);
  const unsigned f = threadIdx.x % Cfg::FPW;
  const unsigned tid = threadIdx.x / Cfg::FPW;
  const cx<T>* __restrict__ tw = static_cast<const cx<T>*>(a.tw);
  const long long ngroups = strided_ngroups<Cfg>(a);
  long long g = blockIdx.x;
  if (g >= ngroups) return;
  static_assert(STW == 0 || LTW == 0, "the modifier sits on one side of the stage");
  strided_copy_twiddles<Cfg>(lds, tw);
  strided_copy_stw<Cfg, STW != 0 ? STW : LTW>(a);
  cx<T> cur[Cfg::bpt(0)][Cfg::Seq::r[0]];
  cx<T> nxt[Cfg::bpt(0)][Cfg::Seq::r[0]];
  bool live, live_n = false;
  long long c0, c0_n = 0, nlive = 0, nlive_n = 0;
  auto io = strided_group<Cfg, SPLIT>(a, g, f, &live, &c0, &nlive);
  auto io_n = io;
  // pass 0 runs on the TIN lane mapping (f0, tid0), the later passes on (f, tid)
  unsigned f0 = f, tid0 = tid;
  bool live0 = live;
  if constexpr (TIN != 0) tin_lanes<Cfg, TIN>(&f0, &tid0, &live0, nlive);
  strided_pass0_load<Cfg, BWD>(io, a, f0, tid0, live0, cur);
  for (; g < ngroups; g += gridDim.x) {
    strided_pass0_compute<Cfg, TIN, LTW>(cur, f0, tid0, lds, &a, c0);
    const long long gn = g + gridDim.x;
    if (gn < ngroups) {
      io_n = strided_group<Cfg, SPLIT>(a, gn, f, &live_n, &c0_n, &nlive_n);
      bool live0_n = live_n;
      if constexpr (TIN != 0) tin_lanes<Cfg, TIN>(&f0, &tid0, &live0_n, nlive_n);
      strided_pass0_load<Cfg, BWD>(io_n, a, f0, tid0, live0_n, nxt);
    }
    strided_passes<Cfg, BWD, STW, 1, decltype(io), false, false, TIN>(io, a, f, tid, live, c0, lds, tw, nlive);
    sfor<0, Cfg::bpt(0)>([&](auto i_) PFA_LAMBDA {
      sfor<0, Cfg::Seq::r[0]>([&](auto t_) PFA_LAMBDA { cur[decltype(i_)::value][decltype(t_)::value] = nxt[decltype(i_)::value][decltype(t_)::value]; });
    });
    io = io_n;
    live = live_n;
    c0 = c0_n;
    nlive = nlive_n;
  }
}

/// Strided kernel with a row-shaped side copied through LDS (see strided_pitch).  Interleaved storage, no store
/// modifier.  ROW_IN: in_stride must be 1 (FFT f starts at f * in_fdist); ROW_OUT: out_stride must be 1.
/// SPLIT 3 (with ROW_IN): the four-step stage B of SPLIT_COMPLEX data -- interleaved rows of the scratch in, the user's
/// two planes out (column-shaped stores of the last pass).  Addressed f-fastest that stage read 8 bytes per lane from
/// FPW different rows (fp32 N = 65536 x 2Ki: 1339 us per GiB); staged it reads whole lines.
/// WALK 1 (the kernels compiled at commit): the groups through strided_group_walk -- the XCD-contiguous walk when the
/// column-shaped side has an unaligned row pitch (P -> BI at a batch count that is no multiple of a line: the output's partial
/// lines of neighbouring groups meet in one L2; fp32 N = 1024 x 131 077 0.233 of the HBM peak streamed, see aux_of_policy)
template <typename Cfg, bool BWD, bool ROW_IN, bool ROW_OUT, int SPLIT = 0, int WALK = 0>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_strided_row_kernel(const strided_args a) {
  using T = typename Cfg::T;
  static_assert(Cfg::NP >= 2 && (ROW_IN || ROW_OUT));
  static_assert(SPLIT == 0 || (SPLIT == 3 && ROW_IN && !ROW_OUT), "row-staged forms: interleaved, or scratch -> planes");
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem_strided);
  constexpr int PITCH = strided_pitch<Cfg, true>();
  constexpr int CH = Cfg::FPW * Cfg::N;
  constexpr int EPT = (CH + Cfg::WG - 1) / Cfg::WG;
  constexpr unsigned ES = sizeof(cx<T>);
  const unsigned f = threadIdx.x % Cfg::FPW;
  const unsigned tid = threadIdx.x / Cfg::FPW;
  const cx<T>* __restrict__ tw = static_cast<const cx<T>*>(a.tw);
  const long long ngroups = strided_ngroups<Cfg>(a);
  // (the body stands twice: the pre-compiled instantiations -- WALK 0 -- are instruction for instruction the kernels of round 5,
  //  which a shared lambda did not guarantee)
  if constexpr (WALK == 0) {
    for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    bool live;
    long long c0;
    long long left;
    const auto io = strided_group<Cfg, SPLIT>(a, g, f, &live, &c0, &left);
    if constexpr (ROW_IN) {
      sfor<0, EPT>([&](auto k_) PFA_LAMBDA {
        constexpr int k = decltype(k_)::value;
        const unsigned e = threadIdx.x + k * Cfg::WG;
        if (CH % Cfg::WG == 0 || e < CH) {
          const unsigned ef = e / Cfg::N, ei = e % Cfg::N;
          const unsigned voff = static_cast<long long>(ef) < left ? (ef * a.in_fdist + ei) * ES : 0xFFFFFFF0u;
          cx<T> x = io.load(voff, 0);
          if constexpr (BWD) x.im = -x.im;
          lds[ei * PITCH + ef] = x;
        }
      });
      __syncthreads();
    }
    strided_passes<Cfg, BWD, false, 0, decltype(io), ROW_IN, ROW_OUT>(io, a, f, tid, live, c0, lds, tw);
    if constexpr (ROW_OUT) {
      const T scale = static_cast<T>(a.scale);
      sfor<0, EPT>([&](auto k_) PFA_LAMBDA {
        constexpr int k = decltype(k_)::value;
        const unsigned e = threadIdx.x + k * Cfg::WG;
        if (CH % Cfg::WG == 0 || e < CH) {
          const unsigned ef = e / Cfg::N, ei = e % Cfg::N;
          cx<T> y = lds[ei * PITCH + ef];
          if constexpr (BWD) y.im = -y.im;
          y.re *= scale;
          y.im *= scale;
          const unsigned voff = static_cast<long long>(ef) < left ? (ef * a.out_fdist + ei) * ES : 0xFFFFFFF0u;
          io.store(y, voff, 0);
        }
      });
      __syncthreads();
    }
    }
  } else {
    strided_group_walk(a, ngroups, [&](long long g) PFA_LAMBDA {
    bool live;
    long long c0;
    long long left;
    const auto io = strided_group<Cfg, SPLIT>(a, g, f, &live, &c0, &left);
    if constexpr (ROW_IN) {
      sfor<0, EPT>([&](auto k_) PFA_LAMBDA {
        constexpr int k = decltype(k_)::value;
        const unsigned e = threadIdx.x + k * Cfg::WG;
        if (CH % Cfg::WG == 0 || e < CH) {
          const unsigned ef = e / Cfg::N, ei = e % Cfg::N;
          const unsigned voff = static_cast<long long>(ef) < left ? (ef * a.in_fdist + ei) * ES : 0xFFFFFFF0u;
          cx<T> x = io.load(voff, 0);
          if constexpr (BWD) x.im = -x.im;
          lds[ei * PITCH + ef] = x;
        }
      });
      __syncthreads();
    }
    strided_passes<Cfg, BWD, false, 0, decltype(io), ROW_IN, ROW_OUT>(io, a, f, tid, live, c0, lds, tw);
    if constexpr (ROW_OUT) {
      const T scale = static_cast<T>(a.scale);
      sfor<0, EPT>([&](auto k_) PFA_LAMBDA {
        constexpr int k = decltype(k_)::value;
        const unsigned e = threadIdx.x + k * Cfg::WG;
        if (CH % Cfg::WG == 0 || e < CH) {
          const unsigned ef = e / Cfg::N, ei = e % Cfg::N;
          cx<T> y = lds[ei * PITCH + ef];
          if constexpr (BWD) y.im = -y.im;
          y.re *= scale;
          y.im *= scale;
          const unsigned voff = static_cast<long long>(ef) < left ? (ef * a.out_fdist + ei) * ES : 0xFFFFFFF0u;
          io.store(y, voff, 0);
        }
      });
      __syncthreads();
    }
    });
  }
}

/// LDS bytes of the row variant (odd pitch)
template <typename Cfg>
constexpr size_t strided_row_lds_bytes() {
  return size_t(Cfg::N) * (Cfg::FPW + 1) * sizeof(cx<typename Cfg::T>);
}

/// LDS bytes of the strided kernel for a wg_cfg (unpadded [element][f] image, then the TWL twiddle copy)
template <typename Cfg>
constexpr size_t strided_lds_bytes() {
  return Cfg::NP > 1 ? (size_t(Cfg::N) * Cfg::FPW + Cfg::TWL_ELEMS) * sizeof(cx<typename Cfg::T>) : 0;
}

/// TWL of a strided kernel: leading tables of at most 16 KiB that still fit the CU's LDS next to the image and do
/// not cost residency below 16 waves (same rule as auto_twl; measured on the C5 column kernel: 4.6 -> 5.0 TB/s)
template <typename T, typename Seq, int WG, int FPW>
constexpr int auto_twl_strided() {
  constexpr long long cu_lds = 160 * 1024;
  const long long base = static_cast<long long>(Seq::n) * FPW * static_cast<long long>(sizeof(cx<T>));
  const long long waves = (WG + 63) / 64;
  const long long before = cu_lds / base;
  for (int k = Seq::count - 1; k >= 1; --k) {
    const long long extra = static_cast<long long>(Seq::tw_off(k + 1)) * static_cast<long long>(sizeof(cx<T>));
    if (base + extra > cu_lds) continue;
    const long long after = cu_lds / (base + extra);
    if (extra <= 16 * 1024 && (after == before || after * waves >= 16)) return k;
  }
  return 0;
}
template <typename T, typename Seq, int WG, int FPW, int OCC, int AUX>
using strided_cfg = wg_cfg<T, Seq, WG, FPW, 0, 0, TW_GLOBAL, OCC, AUX, 0, auto_twl_strided<T, Seq, WG, FPW>()>;

/// copy the leading twiddle tables behind the image once per work-group lifetime
template <typename Cfg>
PFA_DEV void strided_copy_twiddles(cx<typename Cfg::T>* lds, const cx<typename Cfg::T>* __restrict__ tw) {
  if constexpr (Cfg::TWL > 0) {
    cx<typename Cfg::T>* twl = lds + Cfg::N * Cfg::FPW;
    for (int i = threadIdx.x; i < Cfg::TWL_ELEMS; i += Cfg::WG) twl[i] = tw[i];
    __syncthreads();
  }
}

/// The walk of a launch's work-groups over its groups: body(g) for every group g this work-group owns.
/// (a two-tier grid like the headline kernel's was measured here: neutral on C3 / ref65536, 1.2 % slower on C5, and
///  the extra loop state alone cost the fp64 n = 256 row-in / column-out shape 15 % -- plain grid-stride loop)
/// strided_args::pair_xcd 1: groups whose input segments are narrower than a 128-byte line share every line with their
/// neighbour group.  Blocks b and b + 8 run on the same XCD and are dispatched back to back (MI355X_MICROARCH.md,
/// "workgroup dispatch"), so they take groups 2k and 2k + 1: the two halves of a line are then fetched by one XCD at
/// about the same time -- 64-byte segments read at 4.1-4.2 instead of 3.5-3.7 TB/s as a plain copy
/// (tools/probes/seg64_pairing.hip; stores do not gain).  The permutation is a bijection on every aligned run of 16
/// blocks; the host rounds the grid to 16.
/// pair_xcd 2: XCD-contiguous -- the blocks of XCD x (b % 8 == x) walk the x-th eighth of the groups in order, so the
/// groups running side by side on an XCD are neighbours in memory.  For stages whose segments do not start on line
/// boundaries (a row pitch that is no multiple of 128 bytes: 68640 = 104 x 660, 5280-byte rows): every 256-byte segment
/// straddles three lines instead of two and the line it shares with the neighbour group was fetched twice, by two XCDs
/// (profiles/r5_pmc_traffic_ref68640.json: stage A read 1.38 x its input).  Speed only: nothing depends on the mapping.
template <typename Body>
PFA_DEV void strided_group_walk(const strided_args& a, long long ngroups, Body&& body) {
  // (ONE loop for the three walks: the body -- a whole transform -- is instantiated once)
  long long g = blockIdx.x, step = gridDim.x, gend = ngroups;
  if (a.pair_xcd == 2 && gridDim.x >= 8u) {
    const unsigned x = blockIdx.x & 7u;
    step = (gridDim.x - x + 7u) >> 3;  // blocks with this b % 8
    g = ngroups * x / 8 + (blockIdx.x >> 3);
    gend = ngroups * (x + 1) / 8;
  } else if (a.pair_xcd == 1 && (gridDim.x & 15u) == 0u) {
    g = (g & ~15ll) + 2 * (g & 7) + ((g >> 3) & 1);
    gend = (ngroups + 15) & ~15ll;
  }
  for (; g < gend; g += step) {
    if (g >= ngroups) continue;  // (uniform: the ragged last run of a paired grid)
    body(g);
  }
}

/// WALK 0: the grid-stride loop of rounds 1-5 with the XCD pairing of narrow segments (pair_xcd 1) -- every pre-compiled
/// instantiation: the 1024-lane stage-A kernels with the store modifier lost 7-10 % when their loop went through
/// strided_group_walk (same registers, another schedule: g32_20 0.349 -> 0.329, g32_22 0.294 -> 0.279 in the first r6 profiles),
/// so their code stays what it was.  WALK 1: strided_group_walk, i.e. also the XCD-contiguous walk -- the kernels compiled
/// at commit (the lengths with unaligned row pitches are theirs).
template <typename Cfg, bool BWD, int STW, int SPLIT = 0, int TIN = 0, bool BIG = false, int WALK = 0>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_strided_kernel(const strided_args a) {
  static_assert(!BIG || TIN == 0, "big groups: the plain form");
  using T = typename Cfg::T;
  // (a single-pass plan -- one lane per FFT, the reference's WORKITEM tier on strided data -- uses no LDS at all)
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem_strided);
  const unsigned f = threadIdx.x % Cfg::FPW;
  const unsigned tid = threadIdx.x / Cfg::FPW;
  const cx<T>* __restrict__ tw = static_cast<const cx<T>*>(a.tw);
  const long long ngroups = strided_ngroups<Cfg>(a);
  strided_copy_twiddles<Cfg>(lds, tw);
  strided_copy_stw<Cfg, STW>(a);
  if constexpr (WALK == 0) {
    const bool pair = a.pair_xcd != 0 && (gridDim.x & 15u) == 0u;
    long long g0 = blockIdx.x;
    if (pair) g0 = (g0 & ~15ll) + 2 * (g0 & 7) + ((g0 >> 3) & 1);
    const long long gend = pair ? ((ngroups + 15) & ~15ll) : ngroups;
    for (long long g = g0; g < gend; g += gridDim.x) {
      if (g >= ngroups) continue;  // (uniform: the ragged last run of a paired grid)
      bool live;
      long long c0;
      long long nlive;
      const auto io = strided_group<Cfg, SPLIT, BIG>(a, g, f, &live, &c0, &nlive);
      // no barrier needed here: the last pass ends its LDS reads with a barrier before the next group's first write
      strided_passes<Cfg, BWD, STW, 0, decltype(io), false, false, TIN>(io, a, f, tid, live, c0, lds, tw, nlive);
    }
  } else {
    strided_group_walk(a, ngroups, [&](long long g) PFA_LAMBDA {
      bool live;
      long long c0;
      long long nlive;
      const auto io = strided_group<Cfg, SPLIT, BIG>(a, g, f, &live, &c0, &nlive);
      strided_passes<Cfg, BWD, STW, 0, decltype(io), false, false, TIN>(io, a, f, tid, live, c0, lds, tw, nlive);
    });
  }
}

}  // namespace pfa
