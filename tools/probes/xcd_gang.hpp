// PROTOTYPE (tools/tune_xcd.hip -DTUNE_GANG): gang-synchronous form of the XCD-local four-step launch.
// Question (VERDICT r4 "missing" item 2 / "next" item 4): can the intermediate of fp32 2^16 stay in an XCD's 4 MiB L2?
// The ticket-queue kernel (stockham_xcd.hpp) needs rings of 3-12 MiB per XCD.  Here G = n2 / FPW work-groups of ONE XCD
// form a gang (members in arrival order on that XCD's ticket); the gang produces a transform together (member r runs
// stage-A task r, stores into the gang's slot), meets at a gang barrier (a counter in that XCD's L2), and consumes it
// together (member r runs stage-B task r) -- slot lifetime = one task, two slots per gang (the next transform's stage A
// may store while a slow member still reads), 4 gangs x 2 x 512 KiB = 4 MiB per XCD at 2 work-groups per CU ... or ONE slot
// (2 MiB) with a second barrier.  HBM latency is hidden inside the work-group: the next transform's stage-A loads are
// issued before the barrier wait.  Every spin is bounded; a spin that gives up sets gctl[2] and the work-group leaves.
// NOT product code: no recovery, no partial gangs (the grid must give every XCD a multiple of G work-groups).
#pragma once
#include "../../portfft_amd/csrc/stockham_xcd.hpp"

namespace pfa {

constexpr unsigned GANG_MAXG = 32;        // gangs per XCD the control block has room for
constexpr unsigned GANG_SPIN = 1u << 21;  // polls

template <typename CfgA_, typename CfgB_, bool BWD, int STW, int TIN, int OCCX, int WG, int SLOTS>
__global__ __launch_bounds__(WG, OCCX) void gang_fourstep_kernel(const xcd_args x, unsigned* gctl_, const int prefetch) {
  using CfgA = typename xcd_with_aux<CfgA_, XCD_AUX_A>::type;
  using CfgB = typename xcd_with_aux<CfgB_, XCD_AUX_B>::type;
  using T = typename CfgA::T;
  using L = xcd_layout<CfgA, CfgB, WG>;
  static_assert(CfgA::WG == WG && CfgB::WG == WG, "prototype: one group per work-group");
  constexpr unsigned GA = CfgB::N / CfgA::FPW, GB = CfgA::N / CfgB::FPW;
  static_assert(GA == GB, "prototype: square splits");
  constexpr unsigned G = GA;
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  cx<T>* const lds0 = reinterpret_cast<cx<T>*>(pfa_smem_strided);
  unsigned* const s_ctl = reinterpret_cast<unsigned*>(pfa_smem_strided + x.lds_ctl_off);
  const unsigned f = threadIdx.x % CfgA::FPW, tid = threadIdx.x / CfgA::FPW;
  const cx<T>* __restrict__ tw_a = static_cast<const cx<T>*>(x.a.tw);
  const cx<T>* __restrict__ tw_b = static_cast<const cx<T>*>(x.b.tw);
  xcd_gu32* const gctl = (xcd_gu32*)gctl_;
  for (int i = threadIdx.x; i < CfgA::TWL_ELEMS; i += WG) lds0[L::TWL_A + i] = tw_a[i];
  if constexpr (!L::SAME_TW) {
    for (int i = threadIdx.x; i < CfgB::TWL_ELEMS; i += WG) lds0[L::TWL_B + i] = tw_b[i];
  }
  if constexpr (STW == 1) {
    const cx<T>* src = static_cast<const cx<T>*>(x.a.stw_tab);
    const int n = x.a.stw_levels << x.a.stw_lshift;
    for (int i = threadIdx.x; i < n; i += WG) lds0[L::STW + i] = src[i];
  }
  const unsigned q = xcd_id();
  if (threadIdx.x == 0) s_ctl[0] = xcd_add(gctl + 64 + q * 32, 1u);
  __syncthreads();
  const unsigned m = s_ctl[0];
  const unsigned gi = m / G, r = m % G;
  if (gi >= GANG_MAXG) return;
  xcd_gu32* const gw = gctl + 4096 + (q * GANG_MAXG + gi) * 64;  // [0] barrier A, [16] barrier B, [32 + 2 p] claim records
  const long long slot_elems = static_cast<long long>(CfgA::N) * CfgB::N;
  const unsigned batch = static_cast<unsigned>(x.batch);
  // thread 0: wait for *p >= want (bounded); every lane learns the verdict behind the barrier
  auto wait_ge = [&](xcd_gu32* p, unsigned want) PFA_LAMBDA -> bool {
    if (threadIdx.x == 0) {
      unsigned v = xcd_load(p), n = 0;
      while (v < want && n < GANG_SPIN) {
        __builtin_amdgcn_s_sleep(1);
        v = xcd_load(p);
        ++n;
      }
      s_ctl[1] = v >= want ? 1u : 0u;
      if (v < want) xcd_add(gctl + 2, 1u);
    }
    __syncthreads();
    const bool ok = s_ctl[1] != 0u;
    __syncthreads();
    return ok;
  };
  // claim record of iteration `it`: {it + 1, transform + 1 (0: none left)}; member 0 claims, everybody reads
  auto claim = [&](unsigned it) PFA_LAMBDA {
    const unsigned g = xcd_add(gctl + 0, 1u);
    xcd_gu32* rec = gw + 32 + 2 * (it & 3u);
    __hip_atomic_store(rec + 1, g < batch ? g + 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec, it + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto claimed = [&](unsigned it, unsigned* g1) PFA_LAMBDA -> bool {
    if (threadIdx.x == 0) {
      xcd_gu32* rec = gw + 32 + 2 * (it & 3u);
      unsigned tag = xcd_load(rec), n = 0;
      while (tag != it + 1u && n < GANG_SPIN) {
        __builtin_amdgcn_s_sleep(1);
        tag = xcd_load(rec);
        ++n;
      }
      s_ctl[2] = tag == it + 1u ? 1u : 0u;
      s_ctl[3] = xcd_load(rec + 1);
      if (tag != it + 1u) xcd_add(gctl + 2, 1u);
    }
    __syncthreads();
    const bool ok = s_ctl[2] != 0u;
    *g1 = s_ctl[3];
    __syncthreads();
    return ok;
  };
  if (threadIdx.x == 0 && r == 0) {
    claim(0u);
    claim(1u);
  }
  cx<T> cur[CfgA::bpt(0)][CfgA::Seq::r[0]];
  bool have_cur = false;
  unsigned g1 = 0;
  if (!claimed(0u, &g1)) return;
  for (unsigned it = 0; g1 != 0u; ++it) {
    const long long gid = static_cast<long long>(g1) - 1;
    const long long sbase = ((static_cast<long long>(q) * GANG_MAXG + gi) * SLOTS + (it % SLOTS)) * slot_elems;
    if (threadIdx.x == 0 && r == 0) claim(it + 2u);
    bool live;
    long long c0, nlive;
    // ---- stage A, task r
    {
      const auto io = strided_group<CfgA, 0>(x.a, gid * GA + r, f, &live, &c0, &nlive, 0, sbase);
      if (!have_cur) strided_pass0_load<CfgA, BWD>(io, x.a, f, tid, live, cur);
      strided_pass0_compute<CfgA, 0, 0>(cur, f, tid, lds0);
      // one slot: the previous transform's stage B must have been read by every member before this one's stores
      if constexpr (SLOTS == 1) {
        if (it > 0 && !wait_ge(gw + 16, G * it)) return;
      }
      strided_passes_range<CfgA, BWD, STW, 1, CfgA::NP, decltype(io)>(io, x.a, f, tid, live, c0, lds0, tw_a, nlive);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) xcd_add(gw + 0, 1u);
    }
    // ---- the next transform's stage-A input, before the gang barrier
    unsigned g1n = 0;
    if (!claimed(it + 1u, &g1n)) return;
    have_cur = false;
    if (prefetch != 0 && g1n != 0u) {
      bool live_n;
      long long c0_n, nlive_n;
      const auto io_n = strided_group<CfgA, 0>(x.a, (static_cast<long long>(g1n) - 1) * GA + r, f, &live_n, &c0_n, &nlive_n, 0, 0);
      strided_pass0_load<CfgA, BWD>(io_n, x.a, f, tid, live_n, cur);
      have_cur = true;
    }
    // ---- gang barrier: all G stage-A tasks of this transform have stored
    if (!wait_ge(gw + 0, G * (it + 1u))) return;
    // ---- stage B, task r
    {
      const auto io = strided_group<CfgB, 0>(x.b, gid * GB + r, f, &live, &c0, &nlive, sbase, 0);
      unsigned f0 = f, tid0 = tid;
      bool live0 = live;
      if constexpr (TIN != 0) {
        constexpr unsigned TW = tin_width<CfgB, TIN>();
        f0 = (threadIdx.x / TW) % CfgB::FPW;
        tid0 = (threadIdx.x / (TW * CfgB::FPW)) * TW + threadIdx.x % TW;
        live0 = static_cast<long long>(f0) < nlive;
      }
      cx<T> inb[CfgB::bpt(0)][CfgB::Seq::r[0]];
      strided_pass0_load<CfgB, BWD>(io, x.b, f0, tid0, live0, inb);
      strided_pass0_compute<CfgB, TIN, 0>(inb, f0, tid0, lds0);
      if constexpr (SLOTS == 1) {
        if (threadIdx.x == 0) xcd_add(gw + 16, 1u);  // (behind pass 0's barrier: this member's input is in registers)
      }
      strided_passes_range<CfgB, BWD, 0, 1, CfgB::NP, decltype(io), TIN>(io, x.b, f, tid, live, c0, lds0, tw_b, nlive);
    }
    g1 = g1n;
  }
}

}  // namespace pfa
