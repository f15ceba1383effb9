// XCD-local four-step kernel: both stages of an N = n1 x n2 transform in ONE persistent launch, the intermediate of a
// transform written and read by work-groups of the SAME XCD so that it lives in that XCD's 4 MiB L2.
//
// Role in the reference: the GLOBAL tier (/root/reference/src/portfft/dispatcher/global_dispatcher.hpp:343-408) runs one
// kernel per factor and keeps `num_batches_in_l2` transforms in flight so that the intermediate stays in the last-level
// cache (committed_descriptor_impl.hpp:603-611).  The two-launch plan of this library (plan_global.cpp, plan_global) does the
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

/// Bound of a hand-off wait: polls of >= 170 ns of the WAITING wave's own running time (s_sleep + one L2 round trip),
/// i.e. at least 3 s -- a ticket holder that a co-tenant's queues de-schedule for a few time slices (CWSR) must not look
/// like a lost task (round 4 gave up after 45 ms), and a waiter that is itself de-scheduled does not count the time it
/// did not run.  (A wall-clock bound -- s_memrealtime in the loop -- was measured: the two live scalar registers cost the
/// 256-lane stage bodies 0.6 % of the whole launch; profiles/r5_notes.md.)  A wait that does give up costs nothing but
/// time: the launch is recomputed by the recovery launch behind it.
constexpr unsigned XCD_SPIN_LIMIT = 1u << 24;

__device__ __forceinline__ unsigned xcd_load(xcd_gu32* p) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ unsigned xcd_add(xcd_gu32* p, unsigned v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/// The next ticket of a queue.  An atomic INCREMENT (wrapping at 2^32 - 1, i.e. never), not an add: hipcc's atomic
/// optimizer rewrites a fetch-add as one atomic per wave + v_readfirstlane of its result, which waits for the atomic on
/// the spot (0.7-1.6 us on wave 0 of every task: measured); it leaves the increment alone, so the ticket returns behind
/// the task's own loads and is first read in duties().
__device__ __forceinline__ unsigned xcd_take(xcd_gu32* p) {
  return __builtin_amdgcn_atomic_inc32((unsigned*)p, 0xFFFFFFFFu, __ATOMIC_RELAXED, "agent");
}

/// a spin gave up: the first one records what it waited for behind the timeout word (tmo[1..5]; the recovery launch
/// copies them to the plan's host report, the tuner reads them directly)
__device__ __forceinline__ void xcd_give_up(xcd_gu32* tmo, unsigned site, unsigned a, unsigned b, unsigned c, unsigned d) {
  if (__hip_atomic_fetch_add(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && (threadIdx.x % 64u) == 0u) {
    tmo[1] = site;
    tmo[2] = a;
    tmo[3] = b;
    tmo[4] = c;
    tmo[5] = d;
  }
}

/// a wait has run out of polls, or (looked at every 64th poll) the launch has already failed elsewhere
__device__ __forceinline__ bool xcd_wait_expired(xcd_gu32* tmo, unsigned n) {
  return n > XCD_SPIN_LIMIT || (n % 64u == 63u && xcd_load(tmo) != 0u);
}

/// Wave-uniform bounded wait for *p >= want (cumulative counters far below 2^31).  False: the wait gave up -- the
/// caller's task must neither store nor signal (stockham_xcd_fourstep_kernel).
/// Acquire side of a hand-off (the contract is written out in front of the kernel): the counter is polled with relaxed
/// agent-scope atomics; the data it guards is read with sc1 loads that the compiler may not move in front of the poll --
/// the empty asm with a memory clobber is that compiler barrier; the hardware issues a wave's memory operations in order.
__device__ __forceinline__ bool xcd_wait_ge(xcd_gu32* p, unsigned want, xcd_gu32* tmo, unsigned site, unsigned k) {
  if (want == 0) return true;
  unsigned v = xcd_load(p);
  for (unsigned n = 0; v < want; ++n) {
    __builtin_amdgcn_s_sleep(2);
    if (xcd_wait_expired(tmo, n)) {
      xcd_give_up(tmo, site, k, want, v, n);
      return false;
    }
    v = xcd_load(p);
  }
  asm volatile("" ::: "memory");
  return true;
}

typedef unsigned xcd_u4 __attribute__((ext_vector_type(4)));
typedef unsigned xcd_u2 __attribute__((ext_vector_type(2)));

/// A wave whose hand-off wait gave up marks its work-group (control word [10]) and ENDS: it has stored nothing for the
/// task at hand and takes part in nothing any more.  Barriers do not wait for ended waves (s_barrier counts the surviving
/// waves of the group); the last arriver of a stage-A task and thread 0 of a stage-B task look at the mark before they
/// signal, every wave looks at it at the top of its next iteration and leaves.  So a task is either complete -- all
/// waves, all stores, then the signal -- or not signalled: what the recovery launch relies on.  Ending the wave (instead
/// of carrying an "ok" flag through the stage bodies) keeps the verdict out of the stage bodies' registers: the flag
/// cost the 256-lane bodies 1.2 % of the whole launch (profiles/r5_notes.md).
__device__ __forceinline__ void xcd_wave_gives_up(unsigned* s_ctl) {
  __hip_atomic_store(&s_ctl[10], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_endpgm();
}

/// claim-map entry {launch epoch, local transform + 1, claimed transform + 1, 0}: one 16-byte PLAIN store -- the line
/// stays in this XCD's L2, where every reader of the queue finds it ...
__device__ __forceinline__ void xcd_map_store(__amdgpu_buffer_rsrc_t map, unsigned M, unsigned epoch, unsigned k, unsigned gid1) {
  xcd_u4 v;
  v.x = epoch;
  v.y = k + 1u;
  v.z = gid1;
  v.w = 0u;
  __builtin_amdgcn_raw_buffer_store_b128(v, map, (k & (M - 1u)) * 16u, 0, 0);
}
/// ... and one 16-byte load past the L1 (sc1: served by the L2)
__device__ __forceinline__ xcd_u4 xcd_map_load(__amdgpu_buffer_rsrc_t map, unsigned M, unsigned k) {
  return __builtin_amdgcn_raw_buffer_load_b128(map, (k & (M - 1u)) * 16u, 0, 16);
}

/// the claim-map entry of local transform k: spins until its tag shows up; returns claimed transform + 1 (0: none)
__device__ __forceinline__ unsigned xcd_wait_claim(__amdgpu_buffer_rsrc_t map, unsigned M, unsigned epoch, unsigned k,
                                                   xcd_gu32* tmo) {
  unsigned lo = 0;
  for (unsigned n = 0;; ++n) {
    const xcd_u4 v = xcd_map_load(map, M, k);
    const unsigned ep = __builtin_amdgcn_readfirstlane(v.x), tag = __builtin_amdgcn_readfirstlane(v.y);
    lo = __builtin_amdgcn_readfirstlane(v.z);
    if (ep == epoch && tag == k + 1u) break;
    __builtin_amdgcn_s_sleep(2);
    if (xcd_wait_expired(tmo, n)) {
      xcd_give_up(tmo, 3u, k, tag, lo, n);
      lo = 0;
      break;
    }
  }
  asm volatile("" ::: "memory");
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

constexpr unsigned XCD_UNKNOWN = 0xFFFFFFFFu;  // a claim-map entry that was not yet published when it was looked at

/// LDS layout of the launch (elements of cx<T>, from the dynamic base): the image region shared by the two stages --
/// HA stage-A images side by side or HB stage-B images --, stage A's leading twiddle tables, stage B's (one copy when the
/// stages share a configuration), the store-modifier tables, the control words.
template <typename CfgA, typename CfgB, int WG>
struct xcd_layout {
  static constexpr int HA = WG / CfgA::WG, HB = WG / CfgB::WG;  // groups a work-group runs side by side per task
  static constexpr size_t IMG_A = size_t(CfgA::N) * CfgA::FPW, IMG_B = size_t(CfgB::N) * CfgB::FPW;
  static constexpr size_t IMAGE = (HA * IMG_A > HB * IMG_B) ? HA * IMG_A : HB * IMG_B;
  static constexpr bool SAME_TW = CfgA::N == CfgB::N && CfgA::TWL == CfgB::TWL && CfgA::NP == CfgB::NP;  // (same radices: checked below)
  static constexpr size_t TWL_A = IMAGE, TWL_B = SAME_TW ? IMAGE : IMAGE + CfgA::TWL_ELEMS;
  static constexpr size_t STW = TWL_B + CfgB::TWL_ELEMS;
  static constexpr size_t bytes(size_t stw_bytes) {
    return ((STW * sizeof(cx<typename CfgA::T>) + stw_bytes + 15) & ~static_cast<size_t>(15)) + XCD_LDS_CTL_BYTES;
  }
};

/// The last work-group out leaves the control block ready for the next launch: counters only -- tickets, done_a /
/// done_b, the launch-wide words.  The claim maps are left alone (xcd_args.hpp); the epoch makes their entries and the
/// per-transform records of this launch invalid for the next one.  `with_timeout`: the recovery launch's clear.
__device__ __forceinline__ void xcd_clear_ctl(const xcd_args& x, xcd_gu32* ctl, unsigned wg, bool with_timeout) {
  const unsigned qw = xcd_queue_words(x.slots, x.map_log2), mw = 4u << x.map_log2;
  const unsigned words = xcd_ctl_words(x.n_queues, x.slots, x.map_log2);
  for (unsigned i = threadIdx.x; i < words; i += wg) {
    bool clear = i <= XCD_W_EXIT || i == XCD_W_REXIT || (with_timeout && i >= XCD_W_TIMEOUT && i < XCD_W_QUEUES);
    if (i >= XCD_W_QUEUES) {
      const unsigned o = (i - XCD_W_QUEUES) % qw;
      clear = o < 32u || o >= 32u + mw;
    }
    if (clear) __hip_atomic_store(ctl + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) xcd_add(ctl + XCD_W_EPOCH, 1u);
}

/// CfgA / CfgB: the strided configurations of the two stages (lengths n1 = CfgA::N, n2 = CfgB::N); WG: lanes of the
/// launch's work-groups, a multiple of both configurations' -- a task is WG / Cfg::WG groups side by side, each on its own
/// LDS image with its own lanes (the passes' barriers are the work-group's: every lane runs the same pass sequence).
/// fp32 256 x 512 runs 512 lanes: two 256-lane stage-A groups (2 x 32 KiB) or one 512-lane stage-B group (64 KiB) per task.
/// STW: store modifier W_N^(k1 * c) on stage A's stores (1: tables in LDS, 2: global tables).  TIN: stage B's
/// tiled-input form (1 = tiles of its own group width; the two group widths are equal).  FREERUN (tuner only; results
/// are garbage): 1 no claims and no hand-off waits -- the bound of the task structure; 2 no waits; 3 static transform map
/// instead of claims, waits kept.  OCCX: waves per SIMD the register allocator must leave room for.
///
/// Control traffic stays off the tasks' critical path: thread 0 keeps the work-group TWO tickets ahead -- while ticket
/// t is processed it already knows t' (the next one), looks up t' 's claim-map entry and hand-off counter (memory round
/// trips of 0.5-1.6 us under load) BEHIND the task's own pass-0 loads, takes t'' and publishes {t', transform, counter
/// value} in LDS behind pass 0's barrier.  A wave polls for itself only when the published value was not yet sufficient.
/// A stage-A task waits for its slot only in front of its LAST pass (the stores), so a slot's previous occupant may still
/// be read while the task loads and computes.
template <typename CfgA_, typename CfgB_, bool BWD, int STW, int TIN, int FREERUN = 0, int OCCX = CfgA_::OCC,
          int WG = (CfgA_::WG > CfgB_::WG ? CfgA_::WG : CfgB_::WG)>
__global__ __launch_bounds__(WG, OCCX) void stockham_xcd_fourstep_kernel(const xcd_args x) {
  using CfgA = typename xcd_with_aux<CfgA_, XCD_AUX_A>::type;
  using CfgB = typename xcd_with_aux<CfgB_, XCD_AUX_B>::type;
  using T = typename CfgA::T;
  using L = xcd_layout<CfgA, CfgB, WG>;
  static_assert(WG % CfgA::WG == 0 && WG % CfgB::WG == 0 && WG % 64 == 0, "whole groups and whole waves per work-group");
  static_assert(CfgA::FPW == CfgB::FPW, "the intermediate's tiles are FPW wide on both sides");
  static_assert(CfgA::NP >= 2 && CfgB::NP >= 2, "two passes at least (the LDS exchange carries the ticket)");
  constexpr int HA = L::HA, HB = L::HB;
  constexpr unsigned NW = WG / 64;
  static_assert((CfgB::N / CfgA::FPW) % HA == 0 && (CfgA::N / CfgB::FPW) % HB == 0, "whole tasks");
  constexpr unsigned GA = CfgB::N / CfgA::FPW, GB = CfgA::N / CfgB::FPW;  // groups of a transform per stage
  constexpr unsigned TA = GA / HA;  // stage-A tasks of a transform
  constexpr unsigned TB = GB / HB;  // stage-B tasks
  constexpr unsigned TPT = TA + TB;
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  cx<T>* const lds0 = reinterpret_cast<cx<T>*>(pfa_smem_strided);
  // control words behind everything else in the dynamic region (XCD_LDS_CTL_BYTES): [4 p .. 4 p + 2], p = 0 / 1: the
  // record {ticket, transform + 1 or XCD_UNKNOWN, hand-off counter value} of the iteration with parity p; [8] stage-A
  // arrivals of the work-group's waves; [9] "this work-group clears the control block"; [10] sticky: a wave of this
  // work-group gave up on a hand-off wait -- the work-group signals nothing from then on
  unsigned* const s_ctl = reinterpret_cast<unsigned*>(pfa_smem_strided + x.lds_ctl_off);
  // lanes of a group inside the work-group, per stage
  const unsigned half_a = threadIdx.x / CfgA::WG, lane_a = threadIdx.x % CfgA::WG;
  const unsigned half_b = threadIdx.x / CfgB::WG, lane_b = threadIdx.x % CfgB::WG;
  const unsigned f_a = lane_a % CfgA::FPW, tid_a = lane_a / CfgA::FPW;
  const unsigned f_b = lane_b % CfgB::FPW, tid_b = lane_b / CfgB::FPW;
  cx<T>* const lds_a = lds0 + half_a * L::IMG_A;
  cx<T>* const lds_b = lds0 + half_b * L::IMG_B;
  const cx<T>* __restrict__ tw_a = static_cast<const cx<T>*>(x.a.tw);
  const cx<T>* __restrict__ tw_b = static_cast<const cx<T>*>(x.b.tw);
  xcd_gu32* const ctl = (xcd_gu32*)x.ctl;
  xcd_gu32* const tmo = ctl + XCD_W_TIMEOUT;
  const unsigned q = xcd_id();
  const unsigned S = static_cast<unsigned>(x.slots), M = 1u << x.map_log2;
  if (threadIdx.x == 0) {
    s_ctl[8] = 0u;
    s_ctl[9] = 0u;
    s_ctl[10] = 0u;
  }
  // once per work-group lifetime: the leading twiddle tables of both stages and the store-modifier tables into LDS
  // (x.a / x.b carry the offsets: strided_args::twl_lds_off / stw_lds_off)
  for (int i = threadIdx.x; i < CfgA::TWL_ELEMS; i += WG) lds0[L::TWL_A + i] = tw_a[i];
  if constexpr (!L::SAME_TW) {
    for (int i = threadIdx.x; i < CfgB::TWL_ELEMS; i += WG) lds0[L::TWL_B + i] = tw_b[i];
  }
  if constexpr (STW == 1) {
    const cx<T>* src = static_cast<const cx<T>*>(x.a.stw_tab);
    const int n = x.a.stw_levels << x.a.stw_lshift;
    for (int i = threadIdx.x; i < n; i += WG) lds0[L::STW + i] = src[i];
  }
  __syncthreads();
  if (q < static_cast<unsigned>(x.n_queues)) {  // (an id the census did not see: nothing claimed, nothing lost)
    xcd_gu32* const qb = ctl + XCD_W_QUEUES + q * xcd_queue_words(x.slots, x.map_log2);
    xcd_gu32* const ticket = qb;
    const __amdgpu_buffer_rsrc_t map = __builtin_amdgcn_make_buffer_rsrc((void*)(qb + 32), 0, M * 16u, 0x00020000);
    xcd_gu32* const done = qb + 32 + 4 * M;
    const unsigned epoch = xcd_load(ctl + XCD_W_EPOCH);
    const long long slot_elems = static_cast<long long>(CfgA::N) * CfgB::N;
    const unsigned batch = static_cast<unsigned>(x.batch);
    // transform, slot and hand-off counter of a ticket
    auto decode = [&](unsigned tk, unsigned* kl_, unsigned* r_, bool* is_a_, int* k_) PFA_LAMBDA {
      *kl_ = tk / TPT;
      *r_ = tk % TPT;
      *is_a_ = *r_ < TA;
      *k_ = *is_a_ ? static_cast<int>(*kl_) : static_cast<int>(*kl_) - x.lag;
    };
    auto claim = [&](unsigned kc) PFA_LAMBDA {
      const unsigned g = xcd_add(ctl + XCD_W_NEXT, 1u);
      // where the recovery launches would find this transform (xcd_args::tmap); read by another launch only
      if (g < batch && x.tmap != nullptr) {
        xcd_u2 rec;
        rec.x = epoch + 1u;
        rec.y = (q << 28) | (kc + 1u);
        *reinterpret_cast<xcd_u2*>(x.tmap + 2ull * g) = rec;
      }
      xcd_map_store(map, M, epoch, kc, g < batch ? g + 1u : 0u);
    };
    unsigned t_next = 0;  // thread 0: the ticket after the one being processed
    if (threadIdx.x == 0) {
      const unsigned t0 = xcd_take(ticket);
      t_next = xcd_take(ticket);
      s_ctl[0] = t0;
      s_ctl[1] = XCD_UNKNOWN;
      s_ctl[2] = 0u;
      // the queue's very first ticket claims transforms 0 .. lookahead before anybody -- itself included -- waits for them
      if (t0 == 0u) {
        for (unsigned kc = 0; kc <= static_cast<unsigned>(x.lookahead); ++kc) claim(kc);
      }
    }
    __syncthreads();
    unsigned par = 0;
    bool stop = false;
    int left = -1;  // -1: running; otherwise the iterations that remain, this one included (tickets already in hand)
#ifdef PFA_XCD_PROF
    unsigned long long prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    PFA_XCD_STAMP(p_begin);
    bool more = true;
    for (unsigned it = 0; more && it < x.max_iters; ++it) {
      PFA_XCD_STAMP(p_it0);
      // a wave of this work-group has given up (xcd_wave_gives_up): its lanes are missing from every task from now on,
      // nothing this work-group does would count -- the surviving waves leave (each at its own next iteration)
      if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_ctl[10], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0u) break;
      const unsigned t = __builtin_amdgcn_readfirstlane(s_ctl[4 * par]);
      unsigned gid1 = __builtin_amdgcn_readfirstlane(s_ctl[4 * par + 1]);
      const unsigned depv = __builtin_amdgcn_readfirstlane(s_ctl[4 * par + 2]);
      unsigned kl, r;
      bool is_a;
      int k;
      decode(t, &kl, &r, &is_a, &k);
      const bool has_next = left != 1;  // a further iteration follows (its ticket is t_next)
      // ---- thread 0: take the ticket two ahead, look up the next one's claim and counter (used in duties()).  Issued
      // BEHIND the task's own pass-0 loads: a wave's memory operations return in order, and these go to the memory side
      // (0.5-2 us under load) -- in front of the loads they set the floor of wave 0's wait for its data (measured, fp32
      // 256 x 256: 1.90 -> 1.49 ms; tools/probes/xcd_ci.sh).  Keeping the work-group three tickets ahead and issuing
      // them one task early costs more in carried registers than it saves (1.49 -> 1.68 ms).
      unsigned t2 = 0;
      xcd_u4 e1 = {0u, 0u, 0u, 0u};
      unsigned d1 = 0, k1u = 0;
      bool real1 = false;
      auto control_issue = [&]() PFA_LAMBDA {
        if (threadIdx.x == 0 && has_next) {
          if (left < 0) t2 = xcd_take(ticket);
          unsigned kl1, r1;
          bool a1;
          int k1;
          decode(t_next, &kl1, &r1, &a1, &k1);
          real1 = k1 >= 0;
          if (real1 && FREERUN != 1) {
            k1u = static_cast<unsigned>(k1);
            e1 = xcd_map_load(map, M, k1u);
            d1 = __hip_atomic_load(done + (k1u % S) * 64u + (a1 ? 32u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      };
      // Duties of thread 0 in the middle of a task (a barrier follows): publish the next iteration's record, and -- the
      // first ticket of a batch -- claim the transform `lookahead` batches ahead.  Claims of a queue are chained: entry kc
      // is claimed after entry kc - 1 has been published, so "no transform left" is monotone in the local index (the
      // claimer of kc - 1 holds a lower ticket: no deadlock).
      auto duties = [&]() PFA_LAMBDA {
        if (threadIdx.x == 0) {
          if (has_next) {
            unsigned* rec = s_ctl + 4 * (par ^ 1u);
            rec[0] = t_next;
            rec[1] = !real1 ? 0u : (e1.x == epoch && e1.y == k1u + 1u ? e1.z : XCD_UNKNOWN);
            if constexpr (FREERUN == 1 || FREERUN == 3) rec[1] = XCD_UNKNOWN;
            rec[2] = d1;
            t_next = t2;
          }
          if (r == 0 && kl != 0 && (FREERUN == 0 || FREERUN == 2)) {
            const unsigned kc = kl + static_cast<unsigned>(x.lookahead);
            (void)xcd_wait_claim(map, M, epoch, kc - 1u, tmo);
            claim(kc);
          }
        }
      };
      if (k < 0) gid1 = 0u;
      if (k >= 0 && gid1 == XCD_UNKNOWN) {
        if constexpr (FREERUN == 1 || FREERUN == 3) {  // (timing experiments: static transform map)
          const unsigned g = static_cast<unsigned>(k) * static_cast<unsigned>(x.n_queues) + q;
          gid1 = g < batch ? g + 1u : 0u;
        } else {
          gid1 = xcd_wait_claim(map, M, epoch, static_cast<unsigned>(k), tmo);
        }
      }
      stop = stop || (k >= 0 && !is_a && gid1 == 0u);  // the stage-B side has run out: everything real has a lower ticket
      PFA_XCD_STAMP(p_claimed);
      PFA_XCD_ACC(1, p_it0, p_claimed);
      if (k >= 0 && gid1 != 0u) {
        const unsigned slot = static_cast<unsigned>(k) % S, rnd = static_cast<unsigned>(k) / S;
        const long long sbase = (static_cast<long long>(q) * S + slot) * slot_elems;
        const long long gid = static_cast<long long>(gid1) - 1;
        xcd_gu32* const done_a = done + slot * 64u;
        xcd_gu32* const done_b = done_a + 32;
        bool live;
        long long c0, nlive;
        if (is_a) {
          const auto io = strided_group<CfgA, 0>(x.a, gid * GA + r * HA + half_a, f_a, &live, &c0, &nlive, 0, sbase);
          {
            cx<T> cur[CfgA::bpt(0)][CfgA::Seq::r[0]];
            strided_pass0_load<CfgA, BWD>(io, x.a, f_a, tid_a, live, cur);
            control_issue();
            strided_pass0_compute<CfgA, 0, 0>(cur, f_a, tid_a, lds_a);
          }
          PFA_XCD_STAMP(p_p0);
          PFA_XCD_ACC(4, p_claimed, p_p0);
          duties();
          strided_passes_range<CfgA, BWD, STW, 1, CfgA::NP - 1, decltype(io)>(io, x.a, f_a, tid_a, live, c0, lds_a, tw_a, nlive);
          PFA_XCD_STAMP(p_mid);
          PFA_XCD_ACC(5, p_p0, p_mid);
          // the slot's previous occupant must have been read before this task's stores (the last pass); a wave whose
          // wait gave up ends here (xcd_wave_gives_up): it stores nothing, and its work-group signals nothing any more
          if ((FREERUN == 0 || FREERUN == 3) && depv < rnd * GB && !xcd_wait_ge(done_b, rnd * GB, tmo, 1u, static_cast<unsigned>(k))) {
            xcd_wave_gives_up(s_ctl);
          }
          PFA_XCD_STAMP(p_dep);
          PFA_XCD_ACC(2, p_mid, p_dep);
          strided_pass<CfgA, BWD, STW, CfgA::NP - 1, decltype(io)>(io, x.a, f_a, tid_a, live, c0, lds_a, tw_a, nlive);
          PFA_XCD_STAMP(p_st);
          PFA_XCD_ACC(6, p_dep, p_st);
          // every storing wave waits for its stores to reach the L2; the wave that arrives last signals (HA groups)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          PFA_XCD_STAMP(p_dr);
          PFA_XCD_ACC(7, p_st, p_dr);
          PFA_XCD_CNT(8);
          if (threadIdx.x % 64u == 0u) {
            const unsigned old = __hip_atomic_fetch_add(&s_ctl[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old + 1u) % NW == 0u && __hip_atomic_load(&s_ctl[10], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
              xcd_add(done_a, static_cast<unsigned>(HA));
            }
          }
        } else {
          // all of the transform's stage-A groups have stored -- or this wave ends here, having read and stored nothing,
          // and the work-group does not count the task as read
          if ((FREERUN == 0 || FREERUN == 3) && depv < (rnd + 1u) * GA &&
              !xcd_wait_ge(done_a, (rnd + 1u) * GA, tmo, 2u, static_cast<unsigned>(k))) {
            xcd_wave_gives_up(s_ctl);
          }
          PFA_XCD_STAMP(p_dep);
          PFA_XCD_ACC(3, p_claimed, p_dep);
          const auto io = strided_group<CfgB, 0>(x.b, gid * GB + (r - TA) * HB + half_b, f_b, &live, &c0, &nlive, sbase, 0);
          {
            unsigned f0 = f_b, tid0 = tid_b;
            bool live0 = live;
            if constexpr (TIN != 0) {  // lanes element-fastest inside the FPW x TW input tiles (tin_lanes, on the group's own lanes)
              constexpr unsigned TW = tin_width<CfgB, TIN>();
              f0 = (lane_b / TW) % CfgB::FPW;
              tid0 = (lane_b / (TW * CfgB::FPW)) * TW + lane_b % TW;
              live0 = static_cast<long long>(f0) < nlive;
            }
            cx<T> cur[CfgB::bpt(0)][CfgB::Seq::r[0]];
            strided_pass0_load<CfgB, BWD>(io, x.b, f0, tid0, live0, cur);
            control_issue();
            strided_pass0_compute<CfgB, TIN, 0>(cur, f0, tid0, lds_b);
          }
          PFA_XCD_STAMP(p_p0);
          PFA_XCD_ACC(9, p_dep, p_p0);
          // behind pass 0's barrier every wave has its input in registers: the slot is read (HB groups)
          if (threadIdx.x == 0 && __hip_atomic_load(&s_ctl[10], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
            xcd_add(done_b, static_cast<unsigned>(HB));
          }
          duties();
          strided_passes_range<CfgB, BWD, 0, 1, CfgB::NP, decltype(io), TIN>(io, x.b, f_b, tid_b, live, c0, lds_b, tw_b, nlive);
          PFA_XCD_STAMP(p_st);
          PFA_XCD_ACC(10, p_p0, p_st);
          PFA_XCD_CNT(11);
        }
      } else {
        control_issue();
        duties();
        __syncthreads();
        PFA_XCD_CNT(12);
      }
      // the tickets in hand when the end showed up are still processed (they carry claim duties), no further one is taken
      if (left > 0) {
        if (--left == 0) more = false;
      } else if (stop) {
        left = 2;  // two iterations remain: the published next ticket and the one thread 0 took in this iteration
      }
      par ^= 1u;
    }
    // the iteration bound ended the loop with tickets still in hand: their tasks are not done -- say so (site 4)
    if (more && left != 0 && threadIdx.x % 64u == 0u) xcd_give_up(tmo, 4u, s_ctl[0], x.max_iters, 0u, 0u);
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
  // Leave: every atomic of this work-group has completed before it counts itself out; the last one out of a HEALTHY
  // launch clears the control block for the next one.  A launch in which a wait gave up leaves the block as it is: the
  // recovery launch behind it needs the counters, and clears the block itself.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = xcd_add(ctl + XCD_W_EXIT, 1u);
    s_ctl[9] = (old == gridDim.x - 1u && xcd_load(tmo) == 0u) ? 1u : 0u;
  }
  __syncthreads();
  if (s_ctl[9] != 0u) xcd_clear_ctl(x, ctl, WG, false);
}

/// Recovery launch: enqueued behind EVERY stockham_xcd_fourstep_kernel launch, on the same stream, same LDS layout.
/// Behind a healthy launch (timeout word zero) every work-group reads that one word and leaves.  Behind a launch in
/// which a hand-off wait gave up -- tasks were skipped, the output is incomplete -- it recomputes what is missing, so
/// that whatever follows the execute in stream order (the submission's event included) sees a valid result:
/// the reference's contract for the event its compute_* calls return (committed_descriptor.hpp:242-246).
///
/// What a failed launch leaves behind (the invariants stockham_xcd_fourstep_kernel keeps): a task either runs to the
/// end or -- its wait gave up -- stores nothing and signals nothing, so for the transform with record {q, k}
/// (xcd_args::tmap; slot = k % S, round = k / S) the counters of its slot say
///   done_b >= (round + 1) * GB   every stage-B task has run: complete;
///   done_a >= (round + 1) * GA   stage A is complete and the slot still holds the intermediate (a slot is handed to
///                                its next occupant only when done_b is complete): stage B can be run again from it
///                                -- any number of times, it writes the same values;
///   otherwise (or no record)     no stage-B task has run: the user's input of that transform is untouched, also
///                                when input and output are the same buffer.
/// Phases (xcd_args.hpp): XCD_RECOVER_ALL recomputes every incomplete transform from the input (input and output do not
/// alias); aliasing executes run XCD_RECOVER_STAGE_B (stage B from the rings) and then XCD_RECOVER_REST.  A recomputing
/// work-group uses ring slot blockIdx.x as its private intermediate (grid <= queues * slots): no hand-offs, no waits.
/// The last work-group out of the last phase writes the plan's host report and clears the control block.
template <typename CfgA_, typename CfgB_, bool BWD, int STW, int TIN, int OCCX = CfgA_::OCC,
          int WG = (CfgA_::WG > CfgB_::WG ? CfgA_::WG : CfgB_::WG)>
__global__ __launch_bounds__(WG, OCCX) void stockham_xcd_recover_kernel(const xcd_args x, const int phase) {
  xcd_gu32* const ctl = (xcd_gu32*)x.ctl;
  xcd_gu32* const tmo = ctl + XCD_W_TIMEOUT;
  if (xcd_load(tmo) == 0u) return;  // the launch in front was healthy
  using CfgA = typename xcd_with_aux<CfgA_, XCD_AUX_A>::type;
  using CfgB = typename xcd_with_aux<CfgB_, XCD_AUX_B>::type;
  using T = typename CfgA::T;
  using L = xcd_layout<CfgA, CfgB, WG>;
  constexpr int HA = L::HA, HB = L::HB;
  constexpr unsigned GA = CfgB::N / CfgA::FPW, GB = CfgA::N / CfgB::FPW;
  constexpr unsigned TA = GA / HA, TB = GB / HB;
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  cx<T>* const lds0 = reinterpret_cast<cx<T>*>(pfa_smem_strided);
  unsigned* const s_ctl = reinterpret_cast<unsigned*>(pfa_smem_strided + x.lds_ctl_off);
  const unsigned half_a = threadIdx.x / CfgA::WG, lane_a = threadIdx.x % CfgA::WG;
  const unsigned half_b = threadIdx.x / CfgB::WG, lane_b = threadIdx.x % CfgB::WG;
  const unsigned f_a = lane_a % CfgA::FPW, tid_a = lane_a / CfgA::FPW;
  const unsigned f_b = lane_b % CfgB::FPW, tid_b = lane_b / CfgB::FPW;
  cx<T>* const lds_a = lds0 + half_a * L::IMG_A;
  cx<T>* const lds_b = lds0 + half_b * L::IMG_B;
  const cx<T>* __restrict__ tw_a = static_cast<const cx<T>*>(x.a.tw);
  const cx<T>* __restrict__ tw_b = static_cast<const cx<T>*>(x.b.tw);
  for (int i = threadIdx.x; i < CfgA::TWL_ELEMS; i += WG) lds0[L::TWL_A + i] = tw_a[i];
  if constexpr (!L::SAME_TW) {
    for (int i = threadIdx.x; i < CfgB::TWL_ELEMS; i += WG) lds0[L::TWL_B + i] = tw_b[i];
  }
  if constexpr (STW == 1) {
    const cx<T>* src = static_cast<const cx<T>*>(x.a.stw_tab);
    const int n = x.a.stw_levels << x.a.stw_lshift;
    for (int i = threadIdx.x; i < n; i += WG) lds0[L::STW + i] = src[i];
  }
  __syncthreads();
  const unsigned S = static_cast<unsigned>(x.slots);
  const unsigned epoch = xcd_load(ctl + XCD_W_EPOCH);
  const unsigned qw = xcd_queue_words(x.slots, x.map_log2), mw = 4u << x.map_log2;
  const long long slot_elems = static_cast<long long>(CfgA::N) * CfgB::N;
  auto stage_b = [&](long long g, long long sbase) PFA_LAMBDA {
    for (unsigned r = 0; r < TB; ++r) {
      bool live;
      long long c0, nlive;
      const auto io = strided_group<CfgB, 0>(x.b, g * GB + r * HB + half_b, f_b, &live, &c0, &nlive, sbase, 0);
      {
        // pass 0 as in the persistent launch: lanes element-fastest inside the input tiles, on the GROUP's own lanes
        // (strided_pass' own TIN mapping takes the work-group's lane index: right only for one group per work-group)
        unsigned f0 = f_b, tid0 = tid_b;
        bool live0 = live;
        if constexpr (TIN != 0) {
          constexpr unsigned TW = tin_width<CfgB, TIN>();
          f0 = (lane_b / TW) % CfgB::FPW;
          tid0 = (lane_b / (TW * CfgB::FPW)) * TW + lane_b % TW;
          live0 = static_cast<long long>(f0) < nlive;
        }
        cx<T> cur[CfgB::bpt(0)][CfgB::Seq::r[0]];
        strided_pass0_load<CfgB, BWD>(io, x.b, f0, tid0, live0, cur);
        strided_pass0_compute<CfgB, TIN, 0>(cur, f0, tid0, lds_b);
      }
      strided_passes_range<CfgB, BWD, 0, 1, CfgB::NP, decltype(io), TIN>(io, x.b, f_b, tid_b, live, c0, lds_b, tw_b, nlive);
    }
  };
  for (long long g = blockIdx.x; g < x.batch; g += gridDim.x) {
    // 0: stage A not complete (or never claimed), 1: stage A complete, stage B not, 2: complete
    int state = 0;
    long long ring = 0;
    {
      const xcd_u2 rec = *reinterpret_cast<const xcd_u2*>(x.tmap + 2ull * g);
      const unsigned ry = __builtin_amdgcn_readfirstlane(rec.y);
      const unsigned rq = ry >> 28, rk1 = ry & 0x0FFFFFFFu;
      if (static_cast<unsigned>(__builtin_amdgcn_readfirstlane(rec.x)) == epoch + 1u && rk1 != 0u && rq < static_cast<unsigned>(x.n_queues)) {
        const unsigned k = rk1 - 1u, slot = k % S, rnd = k / S;
        xcd_gu32* const done_a = ctl + XCD_W_QUEUES + rq * qw + 32u + mw + slot * 64u;
        const unsigned da = xcd_load(done_a), db = xcd_load(done_a + 32);
        state = db >= (rnd + 1u) * GB ? 2 : (da >= (rnd + 1u) * GA ? 1 : 0);
        ring = (static_cast<long long>(rq) * S + slot) * slot_elems;
      }
    }
    if (phase == XCD_RECOVER_STAGE_B) {
      if (state == 1) stage_b(g, ring);
      continue;
    }
    if (state == 2 || (state == 1 && phase == XCD_RECOVER_REST)) continue;
    const long long mine = static_cast<long long>(blockIdx.x) * slot_elems;
    for (unsigned r = 0; r < TA; ++r) {
      bool live;
      long long c0, nlive;
      const auto io = strided_group<CfgA, 0>(x.a, g * GA + r * HA + half_a, f_a, &live, &c0, &nlive, 0, mine);
      strided_passes<CfgA, BWD, STW, 0, decltype(io)>(io, x.a, f_a, tid_a, live, c0, lds_a, tw_a, nlive);
    }
    // the work-group's own stores have reached the L2 before its sc1 loads ask for them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage_b(g, mine);
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (phase == XCD_RECOVER_STAGE_B) return;  // (the counters stay for the phase behind this one)
  if (threadIdx.x == 0) s_ctl[9] = xcd_add(ctl + XCD_W_REXIT, 1u) == gridDim.x - 1u ? 1u : 0u;
  __syncthreads();
  if (s_ctl[9] == 0u) return;
  if (threadIdx.x == 0 && x.report != nullptr) {
    for (unsigned i = 0; i < 6u; ++i) __hip_atomic_store(x.report + 1 + i, xcd_load(tmo + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(x.report, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  xcd_clear_ctl(x, ctl, WG, true);
}

/// LDS bytes of the launch (xcd_layout) and the table offsets the stage arguments carry
template <typename CfgA, typename CfgB, int WG = (CfgA::WG > CfgB::WG ? CfgA::WG : CfgB::WG)>
constexpr size_t xcd_lds_bytes(size_t stw_bytes) {
  return xcd_layout<CfgA, CfgB, WG>::bytes(stw_bytes);
}

}  // namespace pfa
