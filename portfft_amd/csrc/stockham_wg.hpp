// Work-group tier: one (or several) whole FFT(s) per work-group, Stockham autosort passes through LDS.
//
// Role in the reference: workgroup_impl + wg_dft + the subgroup/workitem tiers below it
// (/root/reference/src/portfft/dispatcher/workgroup_dispatcher.hpp:94-281,
//  /root/reference/src/portfft/common/workgroup.hpp:319-346, common/subgroup.hpp:271-291).
// The reference runs a Bailey 4-step (n x m) with 64-lane work-groups, cross-lane shuffles for the sub-FFTs and a
// scalar transposing store.  This design is different on purpose (MI355X-first):
//   * N = R0*R1*...  (radices up to 16, compile-time) -> one Stockham pass per radix; every pass reads its R inputs
//     at stride N/R (lane-contiguous: conflict-free ds_read_b64/b128, coalesced global loads) and writes the
//     autosorted positions, so the last pass leaves natural order and stores straight to HBM, lane-contiguous.
//   * pass 0 reads HBM directly, the last pass writes HBM directly: data crosses LDS (passes-1) times.
//   * twiddles W_{Ns*R}^{t*q} depend only on (lane, butterfly slot), never on the batch: a persistent work-group
//     loads them once from the HBM table into VGPRs and reuses them for every FFT it processes.
//   * 256-thread work-groups (4 wave64), 16 points per lane at N=4096; LDS = N complex (+pad) per FFT in flight.
#pragma once
#include "butterflies.hpp"

namespace pfa {

template <int... Rs>
struct radix_list {
  static constexpr int count = sizeof...(Rs);
  static constexpr int r[sizeof...(Rs)] = {Rs...};
  static constexpr int n = (Rs * ... * 1);
  /// product of the radices before pass p (the Stockham stride Ns of pass p)
  static constexpr int ns(int p) {
    int s = 1;
    for (int i = 0; i < p; ++i) s *= r[i];
    return s;
  }
  /// offset (in complex elements) of pass p's twiddles in the plan's table; pass 0 has none.
  static constexpr int tw_off(int p) {
    int o = 0;
    for (int i = 1; i < p; ++i) o += ns(i) * (r[i] - 1);
    return o;
  }
  static constexpr int tw_total = tw_off(sizeof...(Rs));
};

enum : int { TW_GLOBAL = 0, TW_REGS = 1 };

/// Compile-time description of one work-group kernel variant.
///  T       float or double
///  Seq     radix_list<...>
///  WG      threads per work-group
///  FPW     FFTs processed concurrently by one work-group (threads per FFT = WG / FPW)
///  PADS    LDS padding period: PADW extra complex elements after every PADS elements (0 disables).  A period equal
///          to the first radix turns the pass-0 scatter (lane stride R0 elements, a multiple of many banks when R0
///          is even) into a stride of R0 + PADW, and -- every later butterfly stride being a multiple of R0 -- keeps
///          all LDS addresses of a butterfly linear in its index (one address VGPR + immediates).
///  TWM     TW_GLOBAL: twiddles re-read from the table (L1/L2 resident) at every use
///          TW_REGS  : twiddles loaded once per work-group lifetime into VGPRs
///  OCC     waves per SIMD to keep resident (bounds the VGPR budget: 512 / OCC)
///  AUX     cache policy bits of the HBM accesses
///  TWL     with TW_GLOBAL: the tables of passes 1..TWL are copied into LDS (behind the FFT images) once per
///          work-group lifetime and read from there; later passes keep reading the global table (L1/L2).  Measured
///          on runtime-specialised fp32 kernels: +8..16 % when <= 16 KiB of table move to LDS.
///  STAGED  1: small lengths -- the group's FPW*N contiguous elements are copied HBM <-> LDS with fully coalesced
///          accesses and every pass works LDS -> LDS (the reference's global2local / local2global staging,
///          common/transfers.hpp:390-443); 0: pass 0 reads HBM and the last pass writes HBM directly
template <typename T_, typename Seq_, int WG_, int FPW_, int PADS_, int PADW_, int TWM_, int OCC_ = 1, int AUX_ = 0,
          int STAGED_ = 0, int TWL_ = 0>
struct wg_cfg {
  using T = T_;
  using Seq = Seq_;
  static constexpr int WG = WG_;
  static constexpr int FPW = FPW_;
  static constexpr int TPF = WG_ / FPW_;
  static constexpr int N = Seq_::n;
  static constexpr int PADS = PADS_;
  static constexpr int PADW = PADW_;
  static constexpr int TWM = TWM_;
  static constexpr int OCC = OCC_;  // minimum waves per SIMD the register allocator must leave room for
  static constexpr int AUX = AUX_;  // cache-policy bits of the HBM accesses (0 default, 2 = nt streaming)
  static constexpr int STAGED = STAGED_;
  static constexpr int NP = Seq_::count;
  static constexpr int pad(int i) { return PADS_ == 0 ? i : i + ((i / PADS_) * PADW_); }
  static constexpr int LDS_PER_FFT = pad(N - 1) + 1 + (PADS_ == 0 ? 0 : PADW_);
  static constexpr int LDS_ELEMS = (NP > 1 || STAGED_) ? LDS_PER_FFT * FPW_ : 0;
  static constexpr int TWL = TWL_;
  static constexpr int TWL_ELEMS = TWL_ > 0 ? Seq_::tw_off(TWL_ + 1) : 0;
  static constexpr size_t LDS_BYTES = size_t(LDS_ELEMS + TWL_ELEMS) * sizeof(cx<T_>);
  /// butterflies each lane performs in pass p
  static constexpr int bpt(int p) { return (N / Seq_::r[p] + TPF - 1) / TPF; }
  static constexpr int twr_off(int p) {
    int o = 0;
    for (int i = 1; i < p; ++i) o += bpt(i) * (Seq_::r[i] - 1);
    return o;
  }
  static constexpr int TWR_TOTAL = twr_off(NP) > 0 ? twr_off(NP) : 1;
};

/// The TWL a kernel should use: the most leading passes whose tables fit 16 KiB of LDS without pushing the CU below
/// 16 resident waves (or below what it had).  Same rule as the runtime planner (jit_planner.cpp); measured in
/// profiles/r1_notes.md.
template <typename T, typename Seq, int WG, int FPW, int PADS, int PADW, int STAGED = 0>
constexpr int auto_twl() {
  using base = wg_cfg<T, Seq, WG, FPW, PADS, PADW, TW_GLOBAL, 1, 0, STAGED, 0>;
  constexpr long long cu_lds = 160 * 1024;
  const long long base_bytes = base::LDS_BYTES > 0 ? static_cast<long long>(base::LDS_BYTES) : 1;
  const long long waves = (WG + 63) / 64;
  const long long before = cu_lds / base_bytes;
  for (int k = Seq::count - 1; k >= 1; --k) {
    const long long extra = static_cast<long long>(Seq::tw_off(k + 1)) * static_cast<long long>(sizeof(cx<T>));
    const long long after = cu_lds / (base_bytes + extra);
    if (extra <= 16 * 1024 && after >= 1 && (after == before || after * waves >= 16)) return k;
  }
  return 0;
}

/// wg_cfg with TW_GLOBAL and the automatic TWL
template <typename T, typename Seq, int WG, int FPW, int PADS, int PADW, int OCC, int AUX, int STAGED = 0>
using wg_cfg_twl =
    wg_cfg<T, Seq, WG, FPW, PADS, PADW, TW_GLOBAL, OCC, AUX, STAGED, auto_twl<T, Seq, WG, FPW, PADS, PADW, STAGED>()>;

template <typename Cfg>
PFA_DEV int lds_pad(int i) {
  if constexpr (Cfg::PADS == 0) {
    return i;
  } else {
    return i + ((i / Cfg::PADS) * Cfg::PADW);
  }
}

/// pad(a + k*unit) == pad(a) + k*pad_step(unit) holds for every k when `unit` is a multiple of the padding period,
/// or when `a` is a multiple of the period and k*unit stays below it (pass 0 of a power-of-two plan).  With a
/// linear step the R LDS accesses of a butterfly share ONE address VGPR and differ only in the immediate offset.
template <typename Cfg>
constexpr bool pad_is_linear(int unit, int count, int a_multiple_of) {
  if (Cfg::PADS == 0) return true;
  const int period = Cfg::PADS;
  if (unit % period == 0) return true;
  return (a_multiple_of % period == 0) && (unit * (count - 1) < period);
}
template <typename Cfg>
constexpr int pad_step(int unit) {
  if (Cfg::PADS == 0) return unit;
  const int period = Cfg::PADS;
  return unit % period == 0 ? unit + (unit / Cfg::PADS) * Cfg::PADW : unit;
}

/// Raw 16- or 8-byte buffer access: one 32-bit lane offset VGPR (voff) serves every access of a butterfly, the
/// butterfly's stride goes into the scalar offset (soff), and the hardware range check drops out-of-range lanes.
using buf_b64_t = decltype(__builtin_amdgcn_raw_buffer_load_b64(declval_of<__amdgpu_buffer_rsrc_t>(), 0u, 0u, 0));
using buf_b128_t = decltype(__builtin_amdgcn_raw_buffer_load_b128(declval_of<__amdgpu_buffer_rsrc_t>(), 0u, 0u, 0));
static_assert(sizeof(buf_b64_t) == 8 && sizeof(buf_b128_t) == 16, "unexpected raw buffer builtin types");

/// The AUX template argument of a kernel configuration carries the cache policy of its HBM accesses: bits 0-7 the
/// policy of the loads (0 default, 2 = nt streaming); bits 8-15, when non-zero, (policy + 1) of the stores, which
/// otherwise follow the loads.  AUX = 2: everything streamed (multi-GiB batches); AUX = 0x102 (0x1102 with sc1, the production writer): streamed loads,
/// default-policy stores -- the writer of an intermediate that should stay in the 256 MiB Infinity Cache;
/// AUX = 0x300: default-policy loads, streamed stores -- its reader (profiles/r2_notes.md).
constexpr int aux_of_loads(int aux) { return aux & 0xFF; }
constexpr int aux_of_stores(int aux) { return (aux >> 8) != 0 ? (aux >> 8) - 1 : (aux & 0xFF); }

template <typename T, int AUX>
PFA_DEV cx<T> buf_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(cx<T>, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, aux_of_loads(AUX)));
  } else {
    return __builtin_bit_cast(cx<T>, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, aux_of_loads(AUX)));
  }
}
template <typename T, int AUX>
PFA_DEV void buf_store(cx<T> v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  if constexpr (sizeof(T) == 4) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(buf_b64_t, v), rsrc, voff, soff, aux_of_stores(AUX));
  } else {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(buf_b128_t, v), rsrc, voff, soff, aux_of_stores(AUX));
    // gfx950 hazard (observed, ROCm 7.2): a >64-bit buffer store with an SGPR soffset still reads its data VGPRs for
    // two more cycles; hipcc only pads the soffset-immediate form, so a VALU write to the data registers right
    // behind the store corrupts the last quad of every 16-lane row.  Pad by hand.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
  }
}

/// Addressing of one work-group's FFTs for the PACKED layout (reference: detail::layout::PACKED,
/// enums.hpp:47-50).  The FPW FFTs of group g are contiguous: the descriptor covers exactly the FFTs of the group
/// that exist (ragged last group: missing FFTs read zeros and their stores are dropped by the range check).
template <typename T, int N, int FPW, int AUX>
struct packed_io {
  static constexpr unsigned ES = sizeof(cx<T>);
  __amdgpu_buffer_rsrc_t rin, rout;
  PFA_DEV packed_io(const cx<T>* in, cx<T>* out, long long g, long long nfft) {
    const long long first = g * FPW;
    const long long left = nfft - first;
    const unsigned bytes = static_cast<unsigned>((left < FPW ? left : FPW) * N * sizeof(cx<T>));
    rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<cx<T>*>(in + first * N), 0, bytes, 0x00020000);
    rout = __builtin_amdgcn_make_buffer_rsrc(out + first * N, 0, bytes, 0x00020000);
  }
  /// byte offset of element j of the group's f-th FFT
  static PFA_DEV unsigned lane_off(unsigned f, unsigned j) { return (f * N + j) * sizeof(cx<T>); }
  /// uniform byte offset of k elements
  static constexpr unsigned step(int k) { return k * sizeof(cx<T>); }
  // the interface the passes use (input and output side may differ: unpacked_io)
  static PFA_DEV unsigned in_off(unsigned f, unsigned j) { return lane_off(f, j); }
  static PFA_DEV unsigned out_off(unsigned f, unsigned j) { return lane_off(f, j); }
  static constexpr unsigned in_step(int k) { return step(k); }
  static constexpr unsigned out_step(int k) { return step(k); }
  /// element e of the group's FPW * N contiguous elements (staged copies)
  static PFA_DEV unsigned in_elem(unsigned e) { return e * ES; }
  static PFA_DEV unsigned out_elem(unsigned e) { return e * ES; }
  PFA_DEV cx<T> load(unsigned voff, unsigned soff) const { return buf_load<T, AUX>(rin, voff, soff); }
  PFA_DEV void store(cx<T> v, unsigned voff, unsigned soff) const { buf_store<T, AUX>(v, rout, voff, soff); }
};

using buf_b32_t = decltype(__builtin_amdgcn_raw_buffer_load_b32(declval_of<__amdgpu_buffer_rsrc_t>(), 0u, 0u, 0));

template <typename T, int AUX>
PFA_DEV T buf_load_scalar(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, aux_of_loads(AUX)));
  } else {
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, aux_of_loads(AUX)));
  }
}
template <typename T, int AUX>
PFA_DEV void buf_store_scalar(T v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  if constexpr (sizeof(T) == 4) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(buf_b32_t, v), rsrc, voff, soff, aux_of_stores(AUX));
  } else {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(buf_b64_t, v), rsrc, voff, soff, aux_of_stores(AUX));
  }
}

/// PACKED layout with SPLIT_COMPLEX storage (reference: complex_storage::SPLIT_COMPLEX, enums.hpp:27; the
/// `else` storage branches of the dispatchers): separate real and imaginary planes, same element indexing.
template <typename T, int N, int FPW, int AUX>
struct packed_split_io {
  static constexpr unsigned ES = sizeof(T);
  __amdgpu_buffer_rsrc_t rin_re, rin_im, rout_re, rout_im;
  PFA_DEV packed_split_io(const T* in_re, const T* in_im, T* out_re, T* out_im, long long g, long long nfft) {
    const long long first = g * FPW;
    const long long left = nfft - first;
    const unsigned bytes = static_cast<unsigned>((left < FPW ? left : FPW) * N * sizeof(T));
    rin_re = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in_re + first * N), 0, bytes, 0x00020000);
    rin_im = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in_im + first * N), 0, bytes, 0x00020000);
    rout_re = __builtin_amdgcn_make_buffer_rsrc(out_re + first * N, 0, bytes, 0x00020000);
    rout_im = __builtin_amdgcn_make_buffer_rsrc(out_im + first * N, 0, bytes, 0x00020000);
  }
  static PFA_DEV unsigned lane_off(unsigned f, unsigned j) { return (f * N + j) * ES; }
  static constexpr unsigned step(int k) { return k * ES; }
  static PFA_DEV unsigned in_off(unsigned f, unsigned j) { return lane_off(f, j); }
  static PFA_DEV unsigned out_off(unsigned f, unsigned j) { return lane_off(f, j); }
  static constexpr unsigned in_step(int k) { return step(k); }
  static constexpr unsigned out_step(int k) { return step(k); }
  static PFA_DEV unsigned in_elem(unsigned e) { return e * ES; }
  static PFA_DEV unsigned out_elem(unsigned e) { return e * ES; }
  PFA_DEV cx<T> load(unsigned voff, unsigned soff) const {
    return {buf_load_scalar<T, AUX>(rin_re, voff, soff), buf_load_scalar<T, AUX>(rin_im, voff, soff)};
  }
  PFA_DEV void store(cx<T> v, unsigned voff, unsigned soff) const {
    buf_store_scalar<T, AUX>(v.re, rout_re, voff, soff);
    buf_store_scalar<T, AUX>(v.im, rout_im, voff, soff);
  }
};

/// UNPACKED layouts (reference: detail::layout::UNPACKED, enums.hpp:47-50; the strided branches of the dispatchers,
/// workitem_dispatcher.hpp:178-204, subgroup_dispatcher.hpp:441-466): element i of transform t at
/// t * dist + i * stride, with dist >= (N - 1) * stride + 1 (transforms do not interleave: rows of a padded
/// matrix, every other sample, ...).  Lanes stay element-fastest, so stride 1 keeps full coalescing.
/// SPLIT: separate real / imaginary planes.
template <typename T, int N, int FPW, int AUX, bool SPLIT>
struct unpacked_io {
  static constexpr unsigned ES = SPLIT ? sizeof(T) : sizeof(cx<T>);
  __amdgpu_buffer_rsrc_t rin, rin_im, rout, rout_im;
  unsigned is, id, os, od;
  PFA_DEV unpacked_io(const void* in, const void* in_im, void* out, void* out_im, long long g, long long nfft,
                      unsigned is_, unsigned id_, unsigned os_, unsigned od_)
      : is(is_), id(id_), os(os_), od(od_) {
    const long long first = g * FPW;
    const long long left = nfft - first;
    const unsigned live = static_cast<unsigned>(left < FPW ? left : FPW);
    const unsigned ibytes = ((live - 1) * id + (N - 1) * is + 1) * ES;
    const unsigned obytes = ((live - 1) * od + (N - 1) * os + 1) * ES;
    const long long ioff = first * static_cast<long long>(id) * ES, ooff = first * static_cast<long long>(od) * ES;
    rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(in)) + ioff, 0, ibytes,
                                            0x00020000);
    rout = __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(out) + ooff, 0, obytes, 0x00020000);
    if constexpr (SPLIT) {
      rin_im = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(in_im)) + ioff, 0, ibytes,
                                                 0x00020000);
      rout_im = __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(out_im) + ooff, 0, obytes, 0x00020000);
    } else {
      rin_im = rin;
      rout_im = rout;
    }
  }
  PFA_DEV unsigned in_off(unsigned f, unsigned j) const { return (f * id + j * is) * ES; }
  PFA_DEV unsigned out_off(unsigned f, unsigned j) const { return (f * od + j * os) * ES; }
  PFA_DEV unsigned in_step(int k) const { return static_cast<unsigned>(k) * is * ES; }
  PFA_DEV unsigned out_step(int k) const { return static_cast<unsigned>(k) * os * ES; }
  PFA_DEV unsigned in_elem(unsigned e) const { return ((e / N) * id + (e % N) * is) * ES; }
  PFA_DEV unsigned out_elem(unsigned e) const { return ((e / N) * od + (e % N) * os) * ES; }
  PFA_DEV cx<T> load(unsigned voff, unsigned soff) const {
    if constexpr (SPLIT) {
      return {buf_load_scalar<T, AUX>(rin, voff, soff), buf_load_scalar<T, AUX>(rin_im, voff, soff)};
    } else {
      return buf_load<T, AUX>(rin, voff, soff);
    }
  }
  PFA_DEV void store(cx<T> v, unsigned voff, unsigned soff) const {
    if constexpr (SPLIT) {
      buf_store_scalar<T, AUX>(v.re, rout, voff, soff);
      buf_store_scalar<T, AUX>(v.im, rout_im, voff, soff);
    } else {
      buf_store<T, AUX>(v, rout, voff, soff);
    }
  }
};

template <typename Cfg, bool BWD, int P, typename IO>
PFA_DEV void wg_pass(const IO& io, unsigned f, cx<typename Cfg::T>* lds, int tid,
                     const cx<typename Cfg::T>* __restrict__ tw, const cx<typename Cfg::T> (&twr)[Cfg::TWR_TOTAL],
                     typename Cfg::T scale) {
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
  constexpr bool from_global = first && !Cfg::STAGED;
  constexpr bool to_global = last && !Cfg::STAGED;

  cx<T> v[BPT][R];
  // ---- gather the R inputs of each butterfly (stride NB: lane-contiguous) ----
  sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    if (!ragged || j < NB) {
      if constexpr (from_global) {
        sfor<0, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          cx<T> x = io.load(io.in_off(f, j), io.in_step(t * NB));
          if constexpr (BWD) x.im = -x.im;
          v[i][t] = x;
        });
      } else if constexpr (pad_is_linear<Cfg>(NB, R, 1)) {
        const cx<T>* p = lds + lds_pad<Cfg>(j);
        sfor<0, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          v[i][t] = p[t * pad_step<Cfg>(NB)];
        });
      } else {
        sfor<0, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          v[i][t] = lds[lds_pad<Cfg>(j + t * NB)];
        });
      }
    }
  });
  // the LDS image is free for the next writer once every lane has its inputs in registers
  if constexpr (!from_global) __syncthreads();
  // ---- twiddle, butterfly, scatter ----
  sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    if (!ragged || j < NB) {
      const unsigned q = j % Ns;
      if constexpr (!first) {
        sfor<1, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          cx<T> w;
          if constexpr (Cfg::TWM == TW_REGS) {
            w = twr[Cfg::twr_off(P) + i * (R - 1) + (t - 1)];
          } else if constexpr (P <= Cfg::TWL) {
            // the LDS copy sits behind the FPW images; `lds` points at this lane's image
            const cx<T>* twl = lds + (Cfg::FPW - f) * Cfg::LDS_PER_FFT;
            w = (twl + Seq::tw_off(P) + (t - 1) * Ns)[q];
          } else {
            w = (tw + Seq::tw_off(P) + (t - 1) * Ns)[q];
          }
          v[i][t] = cmul(v[i][t], w);
        });
      }
      dft<R>(v[i]);
      const unsigned base = (j / Ns) * (Ns * R) + q;
      if constexpr (to_global) {
        sfor<0, R>([&](auto u_) PFA_LAMBDA {
          constexpr int u = decltype(u_)::value;
          cx<T> y = v[i][u];
          if constexpr (BWD) y.im = -y.im;
          y.re *= scale;
          y.im *= scale;
          io.store(y, io.out_off(f, base), io.out_step(u * Ns));
        });
      } else if constexpr (pad_is_linear<Cfg>(Ns, R, Ns * R)) {
        cx<T>* p = lds + lds_pad<Cfg>(base);
        sfor<0, R>([&](auto u_) PFA_LAMBDA {
          constexpr int u = decltype(u_)::value;
          p[u * pad_step<Cfg>(Ns)] = v[i][u];
        });
      } else {
        sfor<0, R>([&](auto u_) PFA_LAMBDA {
          constexpr int u = decltype(u_)::value;
          lds[lds_pad<Cfg>(base + u * Ns)] = v[i][u];
        });
      }
    }
  });
  if constexpr (!to_global) __syncthreads();
}

template <typename Cfg, bool BWD, int P, typename IO>
PFA_DEV void wg_passes(const IO& io, unsigned f, cx<typename Cfg::T>* lds, int tid,
                       const cx<typename Cfg::T>* __restrict__ tw, const cx<typename Cfg::T> (&twr)[Cfg::TWR_TOTAL],
                       typename Cfg::T scale) {
  if constexpr (P < Cfg::NP) {
    wg_pass<Cfg, BWD, P>(io, f, lds, tid, tw, twr, scale);
    wg_passes<Cfg, BWD, P + 1>(io, f, lds, tid, tw, twr, scale);
  }
}

/// Pass 0 split in two for the prefetching kernel: issue the HBM loads of a group ...
template <typename Cfg, bool BWD, typename IO>
PFA_DEV void wg_pass0_load(const IO& io, unsigned f, int tid,
                           cx<typename Cfg::T> (&v)[Cfg::bpt(0)][Cfg::Seq::r[0]]) {
  using T = typename Cfg::T;
  constexpr int R = Cfg::Seq::r[0];
  constexpr int NB = Cfg::N / R;
  constexpr int BPT = Cfg::bpt(0);
  constexpr bool ragged = (NB % Cfg::TPF) != 0;
  sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    if (!ragged || j < NB) {
      sfor<0, R>([&](auto t_) PFA_LAMBDA {
        constexpr int t = decltype(t_)::value;
        cx<T> x = io.load(io.in_off(f, j), io.in_step(t * NB));
        if constexpr (BWD) x.im = -x.im;
        v[i][t] = x;
      });
    }
  });
}

/// ... and butterfly + scatter them into LDS later.
template <typename Cfg>
PFA_DEV void wg_pass0_compute(cx<typename Cfg::T> (&v)[Cfg::bpt(0)][Cfg::Seq::r[0]], cx<typename Cfg::T>* lds,
                              int tid) {
  using T = typename Cfg::T;
  constexpr int R = Cfg::Seq::r[0];
  constexpr int NB = Cfg::N / R;
  constexpr int BPT = Cfg::bpt(0);
  constexpr bool ragged = (NB % Cfg::TPF) != 0;
  sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
    constexpr int i = decltype(i_)::value;
    const unsigned j = tid + i * Cfg::TPF;
    if (!ragged || j < NB) {
      dft<R>(v[i]);
      const unsigned base = j * R;
      if constexpr (pad_is_linear<Cfg>(1, R, R)) {
        cx<T>* p = lds + lds_pad<Cfg>(base);
        sfor<0, R>([&](auto u_) PFA_LAMBDA {
          constexpr int u = decltype(u_)::value;
          p[u * pad_step<Cfg>(1)] = v[i][u];
        });
      } else {
        sfor<0, R>([&](auto u_) PFA_LAMBDA {
          constexpr int u = decltype(u_)::value;
          lds[lds_pad<Cfg>(base + u)] = v[i][u];
        });
      }
    }
  });
  __syncthreads();
}

/// Software-pipelined variant: the HBM loads of the work-group's NEXT group are issued right after pass 0 of the
/// current one, so every work-group keeps loads in flight through its LDS/compute passes instead of alternating
/// between a load phase and a compute phase.  Costs one extra register image of the inputs.
///
/// Two-tier grid (n_main > 0): work-groups [0, n_main) take main_k groups each (group b, b + n_main, ...), the
/// remaining work-groups share the groups behind main_k * n_main a few each.  The hardware hands work-groups to CUs
/// as slots free up, so the many short work-groups at the end balance the launch at a finer grain than main_k groups
/// (fp32 N=4096 x 65536: 12288 x 4 + 8192 x 2 FFTs 686-697 us against 704-705 us for 16384 x 4, profiles/r2_notes.md).
template <typename Cfg, bool BWD>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_wg_prefetch_kernel(
    const cx<typename Cfg::T>* in, cx<typename Cfg::T>* out,
    const cx<typename Cfg::T>* __restrict__ tw, long long nfft, typename Cfg::T scale, long long n_main, int main_k) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  static_assert(Cfg::NP >= 2 && !Cfg::STAGED, "prefetching needs a direct-I/O multi-pass kernel");
  extern __shared__ __attribute__((aligned(16))) char pfa_smem[];
  const int f = threadIdx.x / Cfg::TPF;
  const int tid = threadIdx.x % Cfg::TPF;
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem) + f * Cfg::LDS_PER_FFT;
  using IO = packed_io<T, Cfg::N, Cfg::FPW, Cfg::AUX>;

  cx<T> twr[Cfg::TWR_TOTAL];
  if constexpr (Cfg::TWM == TW_REGS) {
    sfor<1, Cfg::NP>([&](auto p_) PFA_LAMBDA {
      constexpr int p = decltype(p_)::value;
      constexpr int R = Seq::r[p];
      constexpr int Ns = Seq::ns(p);
      sfor<0, Cfg::bpt(p)>([&](auto i_) PFA_LAMBDA {
        constexpr int i = decltype(i_)::value;
        const int q = (tid + i * Cfg::TPF) % Ns;
        sfor<1, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          twr[Cfg::twr_off(p) + i * (R - 1) + (t - 1)] = tw[Seq::tw_off(p) + (t - 1) * Ns + q];
        });
      });
    });
  }

  if constexpr (Cfg::TWL > 0) {  // leading twiddle tables behind the images, once per work-group lifetime
    cx<T>* twl = reinterpret_cast<cx<T>*>(pfa_smem) + Cfg::LDS_ELEMS;
    for (int i = threadIdx.x; i < Cfg::TWL_ELEMS; i += Cfg::WG) twl[i] = tw[i];
    __syncthreads();
  }
  long long ngroups = (nfft + Cfg::FPW - 1) / Cfg::FPW;
  long long g = blockIdx.x, gstride = gridDim.x;
  if (n_main > 0) {
    if (g < n_main) {
      gstride = n_main;
      ngroups = main_k * n_main;  // this work-group's range ends where the tail's begins
    } else {
      gstride = static_cast<long long>(gridDim.x) - n_main;
      g = main_k * n_main + (g - n_main);
    }
  }
  if (g >= ngroups) return;
  cx<T> cur[Cfg::bpt(0)][Seq::r[0]];
  cx<T> nxt[Cfg::bpt(0)][Seq::r[0]];
  {
    const IO io0(in, out, g, nfft);
    wg_pass0_load<Cfg, BWD>(io0, f, tid, cur);
  }
  for (; g < ngroups; g += gstride) {
    const IO io(in, out, g, nfft);
    wg_pass0_compute<Cfg>(cur, lds, tid);
    const long long gn = g + gstride;
    if (gn < ngroups) {
      const IO ion(in, out, gn, nfft);
      wg_pass0_load<Cfg, BWD>(ion, f, tid, nxt);
    }
    const cx<T>* twp = tw;
    if constexpr (Cfg::TWM == TW_GLOBAL) {
      asm volatile("" : "+s"(twp));
    }
    wg_passes<Cfg, BWD, 1>(io, f, lds, tid, twp, twr, scale);
    sfor<0, Cfg::bpt(0)>([&](auto i_) PFA_LAMBDA {
      sfor<0, Seq::r[0]>([&](auto t_) PFA_LAMBDA { cur[decltype(i_)::value][decltype(t_)::value] = nxt[decltype(i_)::value][decltype(t_)::value]; });
    });
  }
}

/// Body shared by the interleaved and the split-storage kernels: `make_io(g)` builds the group's I/O object.
template <typename Cfg, bool BWD, typename MakeIO>
PFA_DEV void stockham_wg_body(MakeIO&& make_io, const cx<typename Cfg::T>* __restrict__ tw, long long nfft,
                              typename Cfg::T scale) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  extern __shared__ __attribute__((aligned(16))) char pfa_smem[];
  const int f = threadIdx.x / Cfg::TPF;
  const int tid = threadIdx.x % Cfg::TPF;
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem) + f * Cfg::LDS_PER_FFT;

  cx<T> twr[Cfg::TWR_TOTAL];
  if constexpr (Cfg::TWM == TW_REGS) {
    sfor<1, Cfg::NP>([&](auto p_) PFA_LAMBDA {
      constexpr int p = decltype(p_)::value;
      constexpr int R = Seq::r[p];
      constexpr int Ns = Seq::ns(p);
      sfor<0, Cfg::bpt(p)>([&](auto i_) PFA_LAMBDA {
        constexpr int i = decltype(i_)::value;
        const int q = (tid + i * Cfg::TPF) % Ns;
        sfor<1, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          twr[Cfg::twr_off(p) + i * (R - 1) + (t - 1)] = tw[Seq::tw_off(p) + (t - 1) * Ns + q];
        });
      });
    });
  }

  if constexpr (Cfg::TWL > 0) {
    cx<T>* twl = reinterpret_cast<cx<T>*>(pfa_smem) + Cfg::LDS_ELEMS;
    for (int i = threadIdx.x; i < Cfg::TWL_ELEMS; i += Cfg::WG) twl[i] = tw[i];
    __syncthreads();
  }
  const long long ngroups = (nfft + Cfg::FPW - 1) / Cfg::FPW;
  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const auto io = make_io(g);
    if constexpr (Cfg::STAGED) {
      // coalesced copy of the group's FPW*N contiguous elements into the (padded) per-FFT LDS images
      constexpr int CH = Cfg::FPW * Cfg::N;
      constexpr int EPT = (CH + Cfg::WG - 1) / Cfg::WG;
      cx<T>* all = reinterpret_cast<cx<T>*>(pfa_smem);
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
    }
    const cx<T>* twp = tw;
    if constexpr (Cfg::TWM == TW_GLOBAL) {
      // keep the table reads inside the loop (L1/L2 hits) instead of letting LICM pin them in VGPRs
      asm volatile("" : "+s"(twp));
    }
    wg_passes<Cfg, BWD, 0>(io, f, lds, tid, twp, twr, scale);
    if constexpr (Cfg::STAGED) {
      constexpr int CH = Cfg::FPW * Cfg::N;
      constexpr int EPT = (CH + Cfg::WG - 1) / Cfg::WG;
      const cx<T>* all = reinterpret_cast<const cx<T>*>(pfa_smem);
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
}

/// Persistent work-group kernel, interleaved complex: work-group g handles FFT groups g, g+G, ...
/// `in` and `out` may be the same buffer (the in-place overloads, N-D passes on the output): the data pointers of the
/// kernels in this file are deliberately NOT __restrict__; only the twiddle tables are.
template <typename Cfg, bool BWD>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_wg_kernel(const cx<typename Cfg::T>* in,
                                                                        cx<typename Cfg::T>* out,
                                                                        const cx<typename Cfg::T>* __restrict__ tw,
                                                                        long long nfft, typename Cfg::T scale) {
  using T = typename Cfg::T;
  stockham_wg_body<Cfg, BWD>(
      [&](long long g) PFA_LAMBDA { return packed_io<T, Cfg::N, Cfg::FPW, Cfg::AUX>(in, out, g, nfft); }, tw, nfft,
      scale);
}

/// Same kernel for SPLIT_COMPLEX storage (separate real / imaginary planes).
template <typename Cfg, bool BWD>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_wg_split_kernel(
    const typename Cfg::T* in_re, const typename Cfg::T* in_im,
    typename Cfg::T* out_re, typename Cfg::T* out_im,
    const cx<typename Cfg::T>* __restrict__ tw, long long nfft, typename Cfg::T scale) {
  using T = typename Cfg::T;
  stockham_wg_body<Cfg, BWD>(
      [&](long long g) PFA_LAMBDA {
        return packed_split_io<T, Cfg::N, Cfg::FPW, Cfg::AUX>(in_re, in_im, out_re, out_im, g, nfft);
      },
      tw, nfft, scale);
}

/// UNPACKED layouts, interleaved or split storage (strides and distances in elements; in_im / out_im only for SPLIT)
template <typename Cfg, bool BWD, bool SPLIT>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_wg_unpacked_kernel(
    const void* in, const void* in_im, void* out, void* out_im,
    const cx<typename Cfg::T>* __restrict__ tw, long long nfft, typename Cfg::T scale, unsigned in_stride,
    unsigned in_dist, unsigned out_stride, unsigned out_dist) {
  using T = typename Cfg::T;
  stockham_wg_body<Cfg, BWD>(
      [&](long long g) PFA_LAMBDA {
        return unpacked_io<T, Cfg::N, Cfg::FPW, Cfg::AUX, SPLIT>(in, in_im, out, out_im, g, nfft, in_stride, in_dist,
                                                                 out_stride, out_dist);
      },
      tw, nfft, scale);
}

}  // namespace pfa
