// XCD-local four-step kernel: both stages of an N = n1 x n2 transform in ONE persistent launch, the intermediate of a
// transform written and read by work-groups of the SAME XCD so that it lives in that XCD's 4 MiB L2.
//
// Role in the reference: the GLOBAL tier (/root/reference/src/portfft/dispatcher/global_dispatcher.hpp:343-408) runs one
// kernel per factor and keeps `num_batches_in_l2` transforms in flight so that the intermediate stays in the last-level
// cache (committed_descriptor_impl.hpp:603-611).  The two-launch plan of this library (plan.cpp, plan_global) does the
// same through the 256 MiB Infinity Cache: every byte crosses the XCD <-> memory fabric four times and a plain copy
// pair of the stages' access shapes tops out at 0.41-0.435 of the HBM peak (profiles/r3_ic_yardstick.txt).  Here a byte
// crosses it twice.  Design (ours, MI355X-specific):
//   * every work-group reads its XCC id (s_getreg HW_REG_XCC_ID -- the hardware's answer, not an assumption about the
//     dispatch order) and joins that XCD's queue; a queue hands out tickets (one returning atomic per task);
//   * ticket t of a queue = task (t % TPT) of "ticket batch" t / TPT: the TA stage-A tasks (one group of FPW columns
//     each) of the queue's local transform kl, then the TB stage-B tasks of local transform kl - lag: by the time a
//     stage-B ticket is taken the stage-A tasks it depends on are `lag` batches old and normally finished, so the
//     dependency poll succeeds at once (no grid barrier, no idle phase);
//   * local transform -> user transform: queues CLAIM transforms from one launch-wide counter, `lookahead` batches
//     ahead (claim map in the control block).  No queue owns a transform statically, so an XCD without resident
//     work-groups loses nothing and the XCDs balance themselves;
//   * the intermediate of local transform k sits in slot k % S of the queue's ring (S transforms of scratch per queue,
//     sized so that the slots in use fit the L2); stage A stores it with PLAIN stores (the lines stay dirty in this
//     XCD's L2, and a slot that is rewritten while resident never leaves the die), stage B reads it with sc1 loads
//     (L1 bypassed, served by the L2 that holds the lines);
//   * hand-offs through cumulative per-slot counters: a stage-A task adds to done_a after every storing wave's
//     `s_waitcnt vmcnt(0)` (the stores have reached the L2), a stage-B task adds to done_b as soon as its input is in
//     registers; stage-A of transform k waits for done_b of k - S (slot drained), stage-B of k for done_a of k.
//     Every wait is for tasks with LOWER tickets of the same queue, which are held by running work-groups: no
//     deadlock whatever the dispatch order or residency; every spin is bounded and reports a timeout word;
//   * the stage bodies are the strided work-group kernels' own passes (stockham_strided.hpp), bit for bit.
// Correctness does not depend on which work-groups share an XCD with which: the reader of a slot is on the writer's
// XCD because both looked up the same physical id.
#pragma once
#include "stockham_strided.hpp"
#include "xcd_args.hpp"

namespace pfa {

typedef __attribute__((address_space(1))) unsigned xcd_gu32;
typedef __attribute__((address_space(1))) unsigned long long xcd_gu64;

__device__ __forceinline__ unsigned xcd_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}

#ifdef PFA_XCD_PROF  // tuner builds: wave 0 of every work-group sums where its cycles go (x.prof: 16 64-bit words)
#define PFA_XCD_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memrealtime()
#define PFA_XCD_ACC(i, a, b) prof_acc[i] += (b) - (a)
#define PFA_XCD_CNT(i) prof_acc[i] += 1
#else
#define PFA_XCD_STAMP(v)
#define PFA_XCD_ACC(i, a, b)
#define PFA_XCD_CNT(i)
#endif

constexpr unsigned XCD_SPIN_LIMIT = 1u << 18;  // polls of ~170 ns: a stuck hand-off gives up after ~45 ms

__device__ __forceinline__ unsigned xcd_load(xcd_gu32* p) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ unsigned xcd_add(xcd_gu32* p, unsigned v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/// a spin gave up: the first one records what it waited for behind the timeout word (tmo[1..7]; read by the tuner and
/// by pfft_plan_check)
__device__ __forceinline__ void xcd_give_up(xcd_gu32* tmo, unsigned site, unsigned a, unsigned b, unsigned c, unsigned d) {
  if (__hip_atomic_fetch_add(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && (threadIdx.x % 64u) == 0u) {
    tmo[1] = site;
    tmo[2] = a;
    tmo[3] = b;
    tmo[4] = c;
    tmo[5] = d;
  }
}

/// wave-uniform bounded wait for *p >= want (cumulative counters far below 2^31)
__device__ __forceinline__ void xcd_wait_ge(xcd_gu32* p, unsigned want, xcd_gu32* tmo, unsigned site, unsigned k) {
  if (want == 0) return;
  unsigned v = xcd_load(p);
  for (unsigned n = 0; v < want; ++n) {
    __builtin_amdgcn_s_sleep(2);
    if (n > XCD_SPIN_LIMIT || (n % 64u == 63u && xcd_load(tmo) != 0u)) {
      xcd_give_up(tmo, site, k, want, v, n);
      break;
    }
    v = xcd_load(p);
  }
}

/// the claim-map entry of local transform k: spins until its tag shows up; returns claimed transform + 1 (0: none)
__device__ __forceinline__ unsigned xcd_wait_claim(xcd_gu64* e, unsigned k, xcd_gu32* tmo) {
  unsigned lo = 0;
  for (unsigned n = 0;; ++n) {
    const unsigned long long v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
    lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
    if (hi == k + 1u) break;
    __builtin_amdgcn_s_sleep(2);
    if (n > XCD_SPIN_LIMIT || (n % 64u == 63u && xcd_load(tmo) != 0u)) {
      xcd_give_up(tmo, 3u, k, hi, lo, n);
      lo = 0;
      break;
    }
  }
  return lo;
}

/// Cache policies of the two stage bodies inside the launch: stage A streams the user's input (nt) and leaves the
/// intermediate dirty in the L2 (plain stores); stage B reads it past the L1 (sc1) and streams the result out (nt).
constexpr int XCD_AUX_A = ((0 + 1) << 8) | 2;
constexpr int XCD_AUX_B = ((2 + 1) << 8) | 16;

template <typename Cfg, int AUX2>
struct xcd_with_aux;
template <typename T, typename Seq, int WG, int FPW, int PADS, int PADW, int TWM, int OCC, int AUX, int STAGED, int TWL,
          int AUX2>
struct xcd_with_aux<wg_cfg<T, Seq, WG, FPW, PADS, PADW, TWM, OCC, AUX, STAGED, TWL>, AUX2> {
  using type = wg_cfg<T, Seq, WG, FPW, PADS, PADW, TWM, OCC, AUX2, STAGED, TWL>;
};

/// CfgA / CfgB: the strided configurations of the two stages (lengths n1 = CfgA::N, n2 = CfgB::N).  This first form
/// takes pairs with equal work-group size, group width and radix sequence (the registered square pairs: fp32 256 x 256,
/// 512 x 512, fp64 256 x 256): one LDS image, one copy of the leading twiddle tables.
/// STW: store modifier W_N^(k1 * c) on stage A's stores (1: tables in LDS, 2: global tables).  TIN: stage B's
/// tiled-input form (1 = tiles of its own group width).
template <typename CfgA_, typename CfgB_, bool BWD, int STW, int TIN>
__global__ __launch_bounds__(CfgA_::WG, CfgA_::OCC) void stockham_xcd_fourstep_kernel(const xcd_args x) {
  using CfgA = typename xcd_with_aux<CfgA_, XCD_AUX_A>::type;
  using CfgB = typename xcd_with_aux<CfgB_, XCD_AUX_B>::type;
  using T = typename CfgA::T;
  static_assert(CfgA::WG == CfgB::WG && CfgA::FPW == CfgB::FPW && CfgA::N == CfgB::N && CfgA::TWL == CfgB::TWL,
                "this form takes square pairs on one configuration");
  static_assert(CfgA::NP >= 2 && CfgB::NP >= 2, "two passes at least (the LDS exchange carries the ticket)");
  static_assert(CfgA::WG % 64 == 0, "whole waves");
  constexpr unsigned NW = CfgA::WG / 64;
  constexpr unsigned TA = CfgB::N / CfgA::FPW;  // stage-A tasks of a transform: n2 columns in groups of FPW
  constexpr unsigned TB = CfgA::N / CfgB::FPW;  // stage-B tasks: n1 rows in groups of FPW
  constexpr unsigned TPT = TA + TB;
  static_assert(CfgB::N % CfgA::FPW == 0 && CfgA::N % CfgB::FPW == 0, "whole groups");
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem_strided);
  // control words behind everything else in the dynamic region: [0], [1] next ticket (double-buffered), [2] stage-A
  // arrivals of the work-group's waves, [3] "this work-group clears the control block"
  unsigned* const s_ctl = reinterpret_cast<unsigned*>(pfa_smem_strided + x.lds_ctl_off);
  const unsigned f = threadIdx.x % CfgA::FPW;
  const unsigned tid = threadIdx.x / CfgA::FPW;
  const cx<T>* __restrict__ tw_a = static_cast<const cx<T>*>(x.a.tw);
  const cx<T>* __restrict__ tw_b = static_cast<const cx<T>*>(x.b.tw);
  xcd_gu32* const ctl = (xcd_gu32*)x.ctl;
  xcd_gu32* const tmo = ctl + XCD_W_TIMEOUT;
  const unsigned q = xcd_id();
  const unsigned S = 1u << x.slots_log2, M = 1u << x.map_log2;
  if (threadIdx.x == 0) {
    s_ctl[2] = 0u;
    s_ctl[3] = 0u;
  }
  strided_copy_twiddles<CfgA>(lds, tw_a);
  strided_copy_stw<CfgA, STW>(x.a);
  if (q < static_cast<unsigned>(x.n_queues)) {  // (an id the census did not see: nothing claimed, nothing lost)
    xcd_gu32* const qb = ctl + XCD_W_QUEUES + q * xcd_queue_words(x.slots_log2, x.map_log2);
    xcd_gu32* const ticket = qb;
    xcd_gu64* const map = (xcd_gu64*)(qb + 32);
    xcd_gu32* const done = qb + 32 + 2 * M;
    const long long slot_elems = static_cast<long long>(CfgA::N) * CfgB::N;
    const unsigned batch = static_cast<unsigned>(x.batch);
    if (threadIdx.x == 0) s_ctl[0] = xcd_add(ticket, 1u);
    __syncthreads();
    unsigned t = __builtin_amdgcn_readfirstlane(s_ctl[0]);
    unsigned par = 1;
    bool stop = false, last = false;
#ifdef PFA_XCD_PROF
    unsigned long long prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    PFA_XCD_STAMP(p_begin);
    for (unsigned it = 0; !last && it < x.max_iters; ++it) {
      PFA_XCD_STAMP(p_it0);
      const unsigned kl = t / TPT, r = t % TPT;
      const bool is_a = r < TA;
      const int k = is_a ? static_cast<int>(kl) : static_cast<int>(kl) - x.lag;
      last = stop;  // the ticket in hand when the end showed up is still processed (it carries a claim duty), no further one taken
      unsigned tn = 0;
      if (threadIdx.x == 0 && !last) tn = xcd_add(ticket, 1u);  // the next ticket: its latency hides behind this task
      // Duties of thread 0 in the middle of a task (a barrier follows): publish the next ticket to the work-group, and --
      // the first ticket of a batch -- claim the transform `lookahead` batches ahead (batch 0: all up to there).  Claims of
      // a queue are chained: entry kc is claimed after entry kc - 1 has been published, so "no transform left" is
      // monotone in the local index (the claimer of kc - 1 holds a lower ticket: no deadlock).
      auto claim = [&](unsigned kc) PFA_LAMBDA {
        const unsigned g = xcd_add(ctl + XCD_W_NEXT, 1u);
        const unsigned long long e = (static_cast<unsigned long long>(kc + 1u) << 32) | (g < batch ? g + 1u : 0u);
        __hip_atomic_store(map + (kc & (M - 1u)), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      auto duties = [&]() PFA_LAMBDA {
        if (threadIdx.x == 0) {
          s_ctl[par] = tn;
          if (r == 0 && kl != 0) {
            const unsigned kc = kl + static_cast<unsigned>(x.lookahead);
            (void)xcd_wait_claim(map + ((kc - 1u) & (M - 1u)), kc - 1u, tmo);
            claim(kc);
          }
        }
      };
      // the queue's very first ticket claims transforms 0 .. lookahead before anybody -- itself included -- waits for them
      if (t == 0 && threadIdx.x == 0) {
        for (unsigned kc = 0; kc <= static_cast<unsigned>(x.lookahead); ++kc) claim(kc);
      }
      unsigned gid1 = 0;
      if (k >= 0) {
        gid1 = xcd_wait_claim(map + (static_cast<unsigned>(k) & (M - 1u)), static_cast<unsigned>(k), tmo);
        stop = stop || (!is_a && gid1 == 0u);  // the stage-B side has run out: everything real has a lower ticket
      }
      PFA_XCD_STAMP(p_claimed);
      PFA_XCD_ACC(1, p_it0, p_claimed);
      if (k >= 0 && gid1 != 0u) {
        const unsigned slot = static_cast<unsigned>(k) & (S - 1u), rnd = static_cast<unsigned>(k) >> x.slots_log2;
        const long long sbase = (static_cast<long long>(q) * S + slot) * slot_elems;
        const long long gid = static_cast<long long>(gid1) - 1;
        xcd_gu32* const done_a = done + slot * 64u;
        xcd_gu32* const done_b = done_a + 32;
        bool live;
        long long c0, nlive;
        if (is_a) {
          xcd_wait_ge(done_b, rnd * TB, tmo, 1u, static_cast<unsigned>(k));  // the slot's previous occupant has been read
          PFA_XCD_STAMP(p_dep);
          PFA_XCD_ACC(2, p_claimed, p_dep);
          const auto io = strided_group<CfgA, 0>(x.a, gid * TA + r, f, &live, &c0, &nlive, 0, sbase);
          strided_pass<CfgA, BWD, STW, 0, decltype(io)>(io, x.a, f, tid, live, c0, lds, tw_a, nlive);
          PFA_XCD_STAMP(p_p0);
          PFA_XCD_ACC(4, p_dep, p_p0);
          duties();
          PFA_XCD_STAMP(p_du);
          PFA_XCD_ACC(5, p_p0, p_du);
          strided_passes<CfgA, BWD, STW, 1, decltype(io)>(io, x.a, f, tid, live, c0, lds, tw_a, nlive);
          PFA_XCD_STAMP(p_st);
          PFA_XCD_ACC(6, p_du, p_st);
          // every storing wave waits for its stores to reach the L2; the wave that arrives last signals
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          PFA_XCD_STAMP(p_dr);
          PFA_XCD_ACC(7, p_st, p_dr);
          PFA_XCD_CNT(8);
          if (threadIdx.x % 64u == 0u) {
            const unsigned old = __hip_atomic_fetch_add(&s_ctl[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old + 1u) % NW == 0u) xcd_add(done_a, 1u);
          }
        } else {
          xcd_wait_ge(done_a, (rnd + 1u) * TA, tmo, 2u, static_cast<unsigned>(k));  // all stage-A tasks have stored
          PFA_XCD_STAMP(p_dep);
          PFA_XCD_ACC(3, p_claimed, p_dep);
          const auto io = strided_group<CfgB, 0>(x.b, gid * TB + (r - TA), f, &live, &c0, &nlive, sbase, 0);
          strided_pass<CfgB, BWD, 0, 0, decltype(io), false, false, TIN>(io, x.b, f, tid, live, c0, lds, tw_b, nlive);
          PFA_XCD_STAMP(p_p0);
          PFA_XCD_ACC(9, p_dep, p_p0);
          // behind pass 0's barrier every wave has its input in registers: the slot is read
          if (threadIdx.x == 0) xcd_add(done_b, 1u);
          duties();
          strided_passes<CfgB, BWD, 0, 1, decltype(io), false, false, TIN>(io, x.b, f, tid, live, c0, lds, tw_b, nlive);
          PFA_XCD_STAMP(p_st);
          PFA_XCD_ACC(10, p_p0, p_st);
          PFA_XCD_CNT(11);
        }
      } else {
        duties();
        __syncthreads();
        PFA_XCD_CNT(12);
      }
      t = __builtin_amdgcn_readfirstlane(s_ctl[par]);
      par ^= 1u;
    }
#ifdef PFA_XCD_PROF
    {
      PFA_XCD_STAMP(p_end);
      PFA_XCD_ACC(0, p_begin, p_end);
      if (threadIdx.x == 0) {
        for (int i = 0; i < 16; ++i) atomicAdd(x.prof + i, prof_acc[i]);
      }
    }
#endif
  }
  // Leave: every atomic of this work-group has completed before it counts itself out; the last one out clears the
  // control block for the next launch (XCD_W_TIMEOUT stays).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = xcd_add(ctl + XCD_W_EXIT, 1u);
    s_ctl[3] = old == gridDim.x - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (s_ctl[3] != 0u) {
    const unsigned words = xcd_ctl_words(x.n_queues, x.slots_log2, x.map_log2);
    for (unsigned i = threadIdx.x; i < words; i += CfgA::WG) {
      if (i < XCD_W_TIMEOUT || i >= XCD_W_QUEUES) __hip_atomic_store(ctl + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

/// LDS bytes of the launch: the configuration's own, the store-modifier tables, 16 bytes of control words
template <typename Cfg>
constexpr size_t xcd_lds_bytes(size_t stw_bytes) {
  return ((strided_lds_bytes<Cfg>() + stw_bytes + 15) & ~static_cast<size_t>(15)) + 16;
}

}  // namespace pfa
