// Runtime specialisation, first half: the PLANNERS -- which radices, lanes, padding, twiddle placement and occupancy a
// length gets (packed, register-resident, strided, rows2d, N-D).  Host-only arithmetic: no hiprtc, no device (the CPU tests
// and tools/jit_plan_dump.cpp / jit_hx_dump.cpp call these directly).  The second half, jit.cpp, compiles, caches and launches
// what is planned here.  Interface: jit.hpp.
#include "jit.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <utility>
#include <functional>
#include <vector>

#include "jit_internal.hpp"

namespace pfa {

namespace {

// ---------------------------------------------------------------------------------------------------------------
// planner
// ---------------------------------------------------------------------------------------------------------------

/// Largest prime radix the runtime-specialised kernels take as one in-register butterfly (PFFT_JIT_MAX_PRIME
/// overrides for experiments).  Primes 37 ... 61 are accepted because a wave64 build of the reference accepts any
/// factor up to its sub-group size (/root/reference/src/portfft/common/subgroup.hpp:226-253).  Measured against the
/// generic tier's "big radix" kernel (tools/perf_primes.py): fp32 N = 37 1.6 -> 5.0 TB/s, 61 1.3 -> 3.3, 37 * 64
/// 1.0 -> 4.3, 59 * 59 0.9 -> 2.3, fp64 37 * 64 0.7 -> 2.9.
static int jit_max_prime() {
  static const int v = jit_knobs::from_env().max_prime;
  return v;
}
#define JIT_MAX_PRIME jit_max_prime()

struct radix_search {
  int precision;
  int cap;  // largest composite radix considered
  std::vector<int> cur, best;
  double best_score = 1e30;
  std::vector<std::pair<double, std::vector<int>>> ranked;  // every factorisation with its score (measured planning)
  bool keep_all = false;
  /// one work-group per CU anyway (the transform takes more than half of the LDS): there is no occupancy left to lose,
  /// and every exchange of such a transform is a full 100+ KiB round trip between barriers -- fewer passes win
  /// (tools/jit_sweep_radices.py: fp32 N = 15625 5^6 2.84 -> 25.25.25 3.34 TB/s, 19683 9.9.9.9.3 3.02 -> 27.27.27 3.36)
  bool single_resident = false;

  // Radices above 16 cost registers (2 VGPRs per element in fp32, 4 in fp64) and, through fewer lanes per FFT,
  // occupancy: measured on MI355X an extra LDS exchange is cheaper (f32 N=243: radices 27,9 run at 2.8 TB/s,
  // 9,9,3 at 5+ TB/s), so they are only taken when a prime factor or the pass limit forces them.
  double penalty(int r) const {
    if (r <= 16) return 0.0;
    if (single_resident && precision != PFFT_PRECISION_F64) return 3.0 * (r - 16);
    return (precision == PFFT_PRECISION_F64 ? 20.0 : 10.0) * (r - 16);
  }

  void consider() {
    double s = 60.0 * static_cast<double>(cur.size());
    int lo = cur[0], hi = cur[0];
    for (int r : cur) {
      s += penalty(r);
      lo = std::min(lo, r);
      hi = std::max(hi, r);
    }
    s += hi - lo;  // balanced radices leave the fewest idle lanes
    if (keep_all) ranked.emplace_back(s, cur);
    if (s < best_score) {
      best_score = s;
      best = cur;
    }
  }

  // non-increasing factorisations of `rem`
  void go(long long rem, int last) {
    if (rem == 1) {
      if (!cur.empty()) consider();
      return;
    }
    if (static_cast<int>(cur.size()) >= MAX_PASSES) return;
    for (int r = std::min<long long>(last, rem); r >= 2; --r) {
      if (rem % r != 0) continue;
      if (r > cap && !(is_prime_i(r) && r <= JIT_MAX_PRIME)) continue;
      cur.push_back(r);
      go(rem / r, r);
      cur.pop_back();
    }
  }
};

std::vector<int> pick_radices(int precision, long long n, int cap, bool single_resident = false) {
  if (n == 1) return {1};
  long long rem = n;
  for (int p = 2; p <= JIT_MAX_PRIME; ++p) {
    while (rem % p == 0) rem /= p;
  }
  if (rem != 1) return {};  // prime factor above JIT_MAX_PRIME
  radix_search s{precision, cap, {}, {}, 1e30, {}, false, false};
  s.single_resident = single_resident;
  s.go(n, JIT_MAX_PRIME + 1);
  return s.best;
}

struct lane_choice {
  int tpf = 0;
  int regs = 0;
  double cost = 1e30;
};

/// lanes per FFT: least wasted lane-slots over all passes, inside the register budget
lane_choice pick_lanes(const std::vector<int>& radices, long long n, int t_lo, int t_hi, int target, int reg_goal,
                       int reg_cap, int fpw_fixed, int wg_cap, int align_lanes = 0) {
  lane_choice best;
  for (int t = std::max(1, t_lo); t <= t_hi; ++t) {
    int regs = 0;
    double slots = 0.0;
    for (int r : radices) {
      const long long nb = n / r;
      const long long bpt = (nb + t - 1) / t;
      regs = std::max<int>(regs, static_cast<int>(bpt) * r);
      slots += static_cast<double>(t) * static_cast<double>(bpt) * r / static_cast<double>(n);
    }
    if (regs > reg_cap) continue;
    const int fpw = fpw_fixed > 0 ? fpw_fixed : std::max(1, 256 / t);
    const int wg = t * fpw;
    if (wg > wg_cap) continue;
    const double waves = static_cast<double>((wg + 63) / 64 * 64) / wg;
    double cost = slots / static_cast<double>(radices.size()) * waves;
    cost *= 1.0 + 0.01 * std::max(0, regs - reg_goal);  // register pressure costs occupancy
    // direct HBM access: the lanes of one FFT read t contiguous elements per instruction; chunks that are not whole
    // 128-byte lines split across waves and lines (N=384: 25 lanes 3.3 TB/s, 32...64 lanes 6.0-6.2)
    // (LDS-staged kernels copy whole groups and do not care: same rule as choose_spec_params' `staged`)
    const long long chunk = n / radices[0];
    const bool direct = align_lanes > 0 && t >= align_lanes && !(t < 64 && chunk % (align_lanes / 2) != 0);
    if (direct && t % align_lanes != 0) cost *= 1.15;
    cost += 1e-4 * std::abs(t - target);                 // ties: stay near the target parallelism
    if (cost < best.cost) {
      best.tpf = t;
      best.regs = regs;
      best.cost = cost;
    }
  }
  return best;
}

}  // namespace

size_t spec_lds_bytes(const wg_params& p) {
  auto pad = [&](int i) { return p.pads == 0 ? i : i + ((i / p.pads) * p.padw); };
  const int per_fft = pad(p.n - 1) + 1 + (p.pads == 0 ? 0 : p.padw);
  const bool uses_lds = p.radices.size() > 1 || p.staged != 0;
  size_t tw = 0, ns = 1;  // tables of passes 1..twl live behind the images
  for (size_t i = 0; i < p.radices.size(); ++i) {
    if (i > 0 && static_cast<int>(i) <= p.twl) tw += ns * static_cast<size_t>(p.radices[i] - 1);
    ns *= static_cast<size_t>(p.radices[i]);
  }
  return ((uses_lds ? static_cast<size_t>(per_fft) * static_cast<size_t>(p.fpw) : 0) + tw) * elem_bytes_of(p.precision);
}

std::vector<std::vector<int>> spec_radix_candidates(int precision, long long n, size_t max_lds, int max_candidates) {
  std::vector<std::vector<int>> out;
  wg_params base;
  if (!choose_spec_params(precision, n, max_lds, &base) || base.radices.size() < 2) return out;
  out.push_back(base.radices);
  const bool f64 = precision == PFFT_PRECISION_F64;
  radix_search s{precision, (f64 && n > 32) ? 16 : 32, {}, {}, 1e30, {}, false, false};
  s.single_resident = static_cast<size_t>(n) * elem_bytes_of(precision) > 80 * 1024;
  s.keep_all = true;
  s.go(n, JIT_MAX_PRIME + 1);
  std::sort(s.ranked.begin(), s.ranked.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  auto add = [&](const std::vector<int>& r) {
    if (static_cast<int>(out.size()) >= max_candidates) return;
    if (std::find(out.begin(), out.end(), r) != out.end()) return;
    wg_params p;
    if (choose_spec_params(precision, n, max_lds, &p, &r) && p.radices == r) out.push_back(r);
  };
  // The best factorisations of the planner's pass count, of one pass fewer and of one more (an exchange less sometimes
  // wins although its radices cost registers, sometimes loses: tools/jit_sweep_radices.py -- fp32 N = 6000: 24.25.10 beats
  // 10.10.10.6 by 14 %, N = 3000: 10.10.10.3 beats 20.15.10), in the orders worth timing: largest radix first (the
  // search's), the two leading radices swapped (the first radix sets the HBM access chunk), smallest first.
  const size_t base_passes = base.radices.size();
  std::vector<std::vector<int>> cls[3];  // [0] one pass fewer, [1] as many, [2] one more
  for (const auto& cand : s.ranked) {
    const size_t np = cand.second.size();
    if (np + 1 < base_passes || np > base_passes + 1) continue;
    cls[np + 1 - base_passes].push_back(cand.second);
  }
  auto add_orders = [&](const std::vector<int>& desc, bool reversed_too) {
    add(desc);
    std::vector<int> r = desc;
    if (r.size() >= 2 && r[0] != r[1]) {
      std::swap(r[0], r[1]);
      add(r);
    }
    if (reversed_too) {
      r = desc;
      std::reverse(r.begin(), r.end());
      add(r);
    }
  };
  for (size_t i = 0; i < cls[1].size() && i < 2; ++i) add_orders(cls[1][i], true);
  for (size_t i = 0; i < cls[0].size() && i < 4; ++i) add_orders(cls[0][i], false);
  for (size_t i = 0; i < cls[2].size() && i < 1; ++i) add_orders(cls[2][i], false);
  return out;
}

bool plan_measure_enabled() { return jit_knobs::from_env().plan_measure; }

bool choose_spec_params(int precision, long long n, size_t max_lds, wg_params* out,
                        const std::vector<int>* forced_radices) {
  const int es = elem_bytes_of(precision);
  const bool f64 = precision == PFFT_PRECISION_F64;
  // one FFT per work-group up to the CU's whole LDS (fp32 N <= 20480): 16807 = 7^5 or 19683 = 3^9 as one HBM pass
  // instead of the two of the four-step tier
  if (n < 2 || static_cast<size_t>(n) * es > max_lds) return false;
  wg_params p;
  p.precision = precision;
  p.n = static_cast<int>(n);
  p.radices = pick_radices(precision, n, (f64 && n > 32) ? 16 : 32, static_cast<size_t>(n) * es > 80 * 1024);
  if (p.radices.empty()) return false;
  // a forced sequence may end in "0, lanes": the lanes per transform of a measured choice (plan_global.cpp measured_radices)
  int forced_tpf = 0;
  std::vector<int> forced;
  if (forced_radices != nullptr) forced = *forced_radices;
  if (forced.size() >= 3 && forced[forced.size() - 2] == 0) {
    forced_tpf = forced.back();
    forced.resize(forced.size() - 2);
  }
  bool forced_ok = false;
  if (!forced.empty() && forced.size() <= static_cast<size_t>(MAX_PASSES)) {
    long long prod = 1;
    bool ok = true;
    for (int r : forced) {
      prod *= r;
      ok = ok && r >= 2 && (r <= 32 || (is_prime_i(r) && r <= JIT_MAX_PRIME));
    }
    if (ok && prod == n) {
      p.radices = forced;
      forced_ok = true;
    }
  }
  if (!forced_ok || forced_tpf < 1 || forced_tpf > 1024) forced_tpf = 0;
  // planner experiments (tools/jit_sweep_radices.py): PFFT_JIT_SPEC_RADICES=n:r0xr1x... forces the radix sequence
  bool env_radices = false;
  const jit_knobs kn = jit_knobs::from_env();
  if (const char* e = kn.spec_radices) {
    long long fn = 0;
    char rad[64] = {0};
    if (std::sscanf(e, "%lld:%63[0-9x]", &fn, rad) == 2 && fn == n) {
      env_radices = true;
      std::vector<int> rs;
      long long prod = 1;
      for (const char* c = rad; *c != '\0';) {
        rs.push_back(std::atoi(c));
        prod *= rs.back();
        while (*c != '\0' && *c != 'x') ++c;
        if (*c == 'x') ++c;
      }
      if (prod == n && rs.size() <= static_cast<size_t>(MAX_PASSES)) p.radices = rs;
    }
  }
  // Lengths with a prime factor P of 37 ... 61 (one O(P^2) butterfly per lane): the prime's pass has only n / P butterflies,
  // and with the usual 12 elements per lane most lanes of a transform sit it out (61 x 16: 16 of 64 lanes busy for 80 % of
  // the kernel's instructions).  Lanes per transform = n / P rounded up to a power of two (16 ... 64), and for n / P <= 32
  // the rest as ONE radix in front of the prime.  Measured (tools/probes/prime_tpf.sh, fp32, fraction of the HBM peak):
  // 976 = 16 x 61 0.27 -> 0.48, 1696 = 32 x 53 0.34 -> 0.51, 2021 = 47 x 43 0.275 -> 0.41, 3481 = 59 x 59 0.33 -> 0.38,
  // 2368 = 37 x 8 x 8 0.59 -> 0.62; prime first / last and 8 ... 128 lanes swept.
  int prime_tpf = 0;
  if (p.radices.size() >= 2 && !env_radices && !forced_ok && !kn.no_prime_lanes) {
    int big = 0;
    for (int r : p.radices) big = std::max(big, r);
    const long long nb = n / std::max(big, 1);
    const int prime_min = kn.prime_lanes_min;  // (experiments: PFFT_PRIME_LANES_MIN)
    if (big >= prime_min && is_prime_i(big) && nb >= 16 && nb <= 64) {  // (beyond 64 the planner's own lanes keep one butterfly per lane)
      int t = 16;
      while (t < nb && t < 64) t *= 2;
      prime_tpf = t;
      std::vector<int> rest;
      bool taken = false;
      for (int r : p.radices) {
        if (r == big && !taken) {
          taken = true;
        } else {
          rest.push_back(r);
        }
      }
      if (nb <= 32) {
        p.radices = {static_cast<int>(nb), big};
      } else {
        p.radices.assign(1, big);
        p.radices.insert(p.radices.end(), rest.begin(), rest.end());
      }
    }
  }
  p.aux = 2;
  p.twm = 0;
  if (p.radices.size() == 1) {
    // one lane per FFT (the reference's WORKITEM tier); odd per-FFT LDS pitch keeps the staged copies conflict-free
    p.wg = 256;
    p.fpw = 256;
    p.staged = 1;
    p.regs = p.n;
    if (p.n % 2 == 0) {
      p.pads = p.n;
      p.padw = 1;
    }
    p.occ = f64 ? (p.n <= 16 ? 2 : 1) : (p.n <= 16 ? 4 : 2);
    if (spec_lds_bytes(p) > 80 * 1024) p.wg = p.fpw = 128;
    *out = p;
    return true;
  }
  // elements per lane: about 12; longer fp32 transforms take 32 to keep WG <= 512
  // (measured over 20 mixed-radix lengths: 8...12 elements per lane are as fast as 16 or faster -- fp64 N=5040 5.1 vs
  //  3.75 TB/s, fp32 N=625 4.2 vs 3.7 -- and 24 is slower; the radices here are mostly <= 12)
  const int e_target = (!f64 && n > 8192) ? 32 : 12;
  const int target = std::max<int>(1, static_cast<int>(n / e_target));
  int max_r = 0;
  for (int r : p.radices) max_r = std::max(max_r, r);
  const int reg_cap = std::max(max_r, e_target + e_target / 2);
  // short transforms: LDS holds 160 KiB / (n * es) FFTs per CU, so each needs >= n * es / 160 lanes for 16 waves
  const int occ_lanes = std::min<int>(64, static_cast<int>((n * es + 159) / 160));
  lane_choice lc = pick_lanes(p.radices, n, std::max({1, target / 2, occ_lanes}),
                              std::min<int>(1024, std::max(2 * target, occ_lanes)), target, e_target, reg_cap, 0, 1024,
                              n > 128 ? 128 / es : 0);
  // (the prime rule, above; a planner's choice of exactly twice as many lanes stays -- 41 x 8 x 8 and 61 x 8 x 6 on 128 lanes
  //  are 7 % faster than on 64, 37 x 8 x 8 5 % slower)
  int force_tpf = (prime_tpf > 0 && !(lc.tpf == 2 * prime_tpf && prime_tpf == 64)) ? prime_tpf : 0;
  if (forced_tpf > 0) force_tpf = forced_tpf;
  if (kn.force_tpf != 0) force_tpf = kn.force_tpf;  // tools/jit_sweep.py (PFFT_JIT_FORCE_TPF): force the lanes per FFT
  {
    const int t = force_tpf;
    if (t > 0 && t <= 1024) {
      lc.tpf = t;
      lc.regs = 0;
      for (int r : p.radices) lc.regs = std::max<int>(lc.regs, static_cast<int>((n / r + t - 1) / t) * r);
    }
  }
  if (lc.tpf == 0) return false;
  p.regs = lc.regs;
  // direct HBM access wants every wave instruction to cover >= 128 contiguous bytes; shorter transforms are copied
  // through LDS (STAGED)
  const long long chunk = n / p.radices[0];  // consecutive elements the lanes of one FFT read per instruction
  p.staged = (lc.tpf * es < 128 || n <= 128 || (lc.tpf < 64 && (chunk * es) % 64 != 0)) ? 1 : 0;
  p.fpw = std::max(1, 256 / lc.tpf);
  p.wg = lc.tpf * p.fpw;
  // <= 40 KiB of LDS per work-group keeps four of them (16 waves) on a CU
  auto fit_fpw = [&]() {
    while (p.fpw > 1 && spec_lds_bytes(p) > std::min<size_t>(max_lds, 40 * 1024)) --p.fpw;
    if (64 % lc.tpf == 0) {  // whole waves
      const int per_wave = 64 / lc.tpf;
      p.fpw = std::max(per_wave, p.fpw / per_wave * per_wave);
    }
    p.wg = lc.tpf * p.fpw;
  };
  fit_fpw();
  // The tables of the leading passes move to LDS (TWL) while they stay within 16 KiB and leave 16 waves resident.
  // Measured on 20 lengths (alternating runs): fp32 +3..17 % everywhere (N=4800 4.35 -> 5.09 TB/s, 3125 3.66 -> 4.17,
  // 10080 3.76 -> 4.18), fp64 +1..3 %; whole 64 KiB tables in LDS lose 10-20 % (occupancy).
  {
    const size_t limit = 16 * 1024;
    const size_t base = spec_lds_bytes(p);
    for (int k = static_cast<int>(p.radices.size()) - 1; k >= 1; --k) {
      wg_params q = p;
      q.twl = k;
      const size_t total = spec_lds_bytes(q);
      // the copy may cost resident work-groups only while 16 waves stay on the CU
      const size_t cu_lds = 160 * 1024, waves = static_cast<size_t>((p.wg + 63) / 64);
      const size_t before = cu_lds / std::max<size_t>(base, 1), after = cu_lds / total;
      if (total - base <= limit && total <= max_lds && (after == before || after * waves >= 16)) {
        p.twl = k;
        fit_fpw();
        break;
      }
    }
  }
  // LDS padding (+1 element per 16) only when the first radix is a power of two >= 8: then the pass-0 scatter has a
  // lane stride of 64 or 128 bytes and the padding keeps every later access linear.  Measured (tools/jit_sweep.py,
  // profiles/r1_notes.md): N=1280 (16.10.8) +5 %, 1920 (16.12.10) +4-10 %, 384 (8.8.6) +3 %; with other first
  // radices the same padding costs 10-35 % (N=3000, 5120, 6000: non-linear addresses), and a period equal to the
  // first radix changes nothing.
  if (const char* e = kn.force_pad) {  // tools/jit_sweep.py (PFFT_JIT_FORCE_PAD): force the padding period
    p.pads = std::atoi(e);
    p.padw = p.pads > 0 ? 1 : 0;
    fit_fpw();
  } else if (const int r0 = p.radices[0]; r0 >= 8 && (r0 & (r0 - 1)) == 0 && !p.staged) {
    p.pads = 16;
    p.padw = 1;
    fit_fpw();
    if (spec_lds_bytes(p) > max_lds) p.pads = p.padw = 0;
  }
  if (spec_lds_bytes(p) > max_lds) return false;
  if (f64) {
    p.occ = p.regs <= 16 ? 2 : 1;
  } else {
    p.occ = p.regs <= 16 ? 4 : (p.regs <= 24 ? 3 : 2);
  }
  // (register-resident twiddles, TW_REGS, were measured on the specialised lengths: no gain on most, large losses
  //  on 4-5 pass lengths -- they stay a hand-tuned option of the pre-compiled kernels)
  *out = p;
  return true;
}

/// transforms of more than this many bytes take the register-resident form when it plans (default: what does not fit
/// the LDS; PFFT_JIT_HX_MIN_KIB: experiments with the lengths below, tools/perf_hx.py)
static size_t hx_min_bytes(const jit_knobs& kn, size_t max_lds, int precision) {
  if (kn.hx_min_kib >= 0) return static_cast<size_t>(kn.hx_min_kib) << 10;
  // Below the LDS limit the two forms are within +-5 % of each other (tools/perf_hx_below.py: fp32 8000 ... 19683, fp64
  // 5000 ... 10240) -- except at the very top in fp32, where the LDS-resident kernel has the CU's whole LDS and nothing
  // else: 20480 0.425 -> 0.52 of the HBM peak, 19683 0.43 -> 0.45
  return precision == PFFT_PRECISION_F32 ? std::min<size_t>(max_lds, 152 * 1024) : max_lds;
}

/// transforms of more than this many bytes (and no more than hx_min_bytes) are planned as two register-resident
/// work-groups per CU (PFFT_JIT_HX_PAIRS=0: never; PFFT_JIT_HX_PAIR_MIN_KIB: experiments, tools/perf_hx_pairs.py).
/// Default 80 KiB: up to there the LDS-resident kernel has two work-groups per CU itself.
static size_t hx_pair_min_bytes(const jit_knobs& kn) { return static_cast<size_t>(kn.hx_pair_min_kib) << 10; }

size_t hx_lds_bytes(const wg_params& p) {
  auto pad = [&](int i) { return p.pads == 0 ? i : i + ((i / p.pads) * p.padw); };
  int h = 0;
  for (size_t i = 1; i < p.radices.size(); ++i) h = std::max(h, ((p.radices[i] + 1) / 2) * (p.n / p.radices[i]));
  const int image = pad(h - 1) + 1 + (p.pads == 0 ? 0 : p.padw);
  size_t tw = 0, ns = 1;
  for (size_t i = 0; i < p.radices.size(); ++i) {
    if (i > 0 && static_cast<int>(i) <= p.twl) tw += ns * static_cast<size_t>(p.radices[i] - 1);
    ns *= static_cast<size_t>(p.radices[i]);
  }
  return (static_cast<size_t>(image) + tw) * elem_bytes_of(p.precision);
}

bool choose_hx_params(int precision, long long n, size_t max_lds, wg_params* out, int skip_pairs) {
  const jit_knobs kn = jit_knobs::from_env();
  const int es = elem_bytes_of(precision);
  const bool f64 = precision == PFFT_PRECISION_F64;
  if (n < 1024 || static_cast<size_t>(n) * es > 3 * max_lds) return false;
  // below the threshold of the one-work-group form only the two-work-groups-per-CU plans are looked at
  const bool pairs_only = static_cast<size_t>(n) * es <= hx_min_bytes(kn, max_lds, precision);
  if (pairs_only && (!kn.hx_pairs || static_cast<size_t>(n) * es <= hx_pair_min_bytes(kn))) return false;
  // every factorisation with radices up to 32 (primes included), ranked by the packed planner's score with the "one
  // work-group per CU anyway" weights: fewer passes first, radices above 16 cheap
  radix_search s{precision, 32, {}, {}, 1e30, {}, false, false};
  s.single_resident = true;
  s.keep_all = true;
  long long rem = n;
  for (int p = 2; p <= 31; ++p) {
    while (rem % p == 0) rem /= p;
  }
  if (rem != 1) return false;
  s.go(n, 33);
  std::sort(s.ranked.begin(), s.ranked.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  wg_params best;
  double best_cost = 1e30;
  std::vector<std::pair<double, wg_params>> pair_plans;  // skip_pairs: the k-th best of them
  size_t looked = 0;
  for (const auto& cand : s.ranked) {
    if (cand.second.size() < 2 || cand.second.size() > 4) continue;
    if (++looked > 200) break;
    // the search yields non-increasing sequences; the LARGEST radix goes first: pass 0 is the one pass whose inputs are
    // not read from the image (the image holds ceil(R / 2) / R of the transform for the other radices, exactly half when
    // they are even)
    std::vector<int> rad = cand.second;
    // lanes > 0: one work-group per CU; lanes < 0: TWO work-groups of -lanes lanes per CU, each with half of the LDS and of
    // the registers -- for transforms of up to ~150 KiB (fp32 16384: 0.66 LDS-resident -> 0.7x, tools/tune.hip case 16388)
    for (int lanes_signed : {1024, 896, 768, 640, 512, -512, -448, -384, -320, -256}) {
      const bool pair = lanes_signed < 0;
      const int lanes = pair ? -lanes_signed : lanes_signed;
      if ((pair && !kn.hx_pairs) || (!pair && pairs_only)) continue;
      // Pairs: three passes behind a first radix of 15 and up whose scatter is conflict-free -- an odd radix as it is, an even
      // one with the image padded by one element per R0 (lane stride R0 + 1).  tools/perf_hx_pairs.py, pair against the
      // LDS-resident plan: 32.24.16 (12288) +20 %, 32.27.16 +29 %, 32.30.16 (15360) +32 %, 32.32.16 +14 %, fp64 32.24.8 +28 %;
      // UNPADDED even radices lose: 24.24.16 -24 %, 30.25.16 -15 %, 30.30.16 -7 %, fp64 28.16.16 / 30.16.16 -8 ... -11 %, padded they
      // gain like the powers of two (profiles/r5_perf_hx_pad_even.txt: 30.30.15 -9 % -> +26 %, 28.25.20 +10 % -> +21 %, fp64
      // 30.25.9 -6 % -> +20 %); every four-pass plan on 256 lanes x 72-75 values loses 19 ... 32 %
      if (pair && (rad.size() > 3 || rad[0] < 15)) continue;
      const size_t lds_limit = pair ? (std::min<size_t>(max_lds, 160 * 1024) / 2 - 512) : max_lds;
      const int waves_per_simd = ((pair ? 2 : 1) * (lanes / 64) + 3) / 4;
      const int budget = (512 / waves_per_simd) / 8 * 8;             // VGPRs a wave may allocate
      const int regs_per_elem = f64 ? 4 : 2;
      // twiddles in flight, butterfly temporaries, addresses -- and the slack the allocator needs: fp32 40960 on 512 lanes
      // (80 values per lane, 160 of 256 registers) ran at 0.30 of the HBM peak against 0.35 for its four-step plan, 36864
      // (72 per lane) at 0.43 against 0.37 (tools/perf_hx.py)
      const int overhead = f64 ? 100 : (waves_per_simd >= 4 ? 62 : 100);
      int regs = 0;
      double slots = 0.0;
      for (int r : rad) {
        const long long nb = n / r;
        const long long bpt = (nb + lanes - 1) / lanes;
        regs = std::max<int>(regs, static_cast<int>(bpt) * r);
        slots += static_cast<double>(lanes) * static_cast<double>(bpt) * r / static_cast<double>(n);
      }
      if (regs * regs_per_elem + overhead > budget) continue;
      wg_params p;
      // (the software-pipelined form of the kernel -- PF, stockham_wg_hx.hpp -- is used by the registered fp64 8192
      // entry only: planned here for 512 lanes it gained nothing at any length, profiles/r5_perf_hx_below_pf512.txt)
      p.precision = precision;
      p.n = static_cast<int>(n);
      p.radices = rad;
      p.wg = lanes;
      p.fpw = 1;
      p.twm = 0;
      p.aux = 2;
      p.staged = 0;
      p.regs = regs;
      p.occ = waves_per_simd;
      p.hx_pair = pair ? 1 : 0;
      // (the packed planner's padding rule; for a pair every even first radix -- one work-group per CU showed -4 ... +2 % with it)
      if (const int r0 = rad[0]; r0 >= 8 && ((r0 & (r0 - 1)) == 0 || (pair && r0 % 2 == 0))) {
        p.pads = r0;
        p.padw = 1;
      }
      // the tables of the leading passes in LDS while they stay within 16 KiB and the image still fits
      for (int k = static_cast<int>(rad.size()) - 1; k >= 0; --k) {
        wg_params q = p;
        q.twl = k;
        const size_t total = hx_lds_bytes(q), base = hx_lds_bytes(p);
        if (total - base <= 16 * 1024 && total <= lds_limit) {
          p.twl = k;
          break;
        }
      }
      if (hx_lds_bytes(p) > lds_limit) {
        p.pads = p.padw = 0;
        if (hx_lds_bytes(p) > lds_limit) continue;
      }
      // cost: idle lane-slots of the ragged passes, an exchange per pass, 16 waves hide latency better than 8
      // (fp32 32768: 1024 lanes 5.19 TB/s, 512 lanes 4.55; four passes 4.0-4.6 -- tools/tune.hip case 32768)
      const double idle = slots / static_cast<double>(rad.size()) - 1.0;
      double cost = (1.0 + 0.2 * (static_cast<double>(rad.size()) - 3.0)) * (1.0 + 0.5 * idle);
      cost *= 1.0 + 0.05 * (4 - waves_per_simd);
      if (pair) cost *= 0.8;  // two work-groups overlap each other's HBM phases
      // (a power-of-two first radix first: its butterfly is the cheapest in registers -- fp64 7680 as 30.16.16 needed scratch at
      //  the pair's 256 VGPRs, as 32.16.15 it runs 31 % above the LDS-resident plan)
      if (pair && (rad[0] & (rad[0] - 1)) != 0) cost *= 1.15;
      if (const char* e = kn.hx_force) {  // experiments (tools/perf_hx.py, PFFT_JIT_HX_FORCE): "lanes:r0xr1x..." or "lanes"
        int fl = 0;
        char rs[64] = {0};
        const int got = std::sscanf(e, "%d:%63[0-9x]", &fl, rs);
        std::string mine;
        for (size_t i = 0; i < rad.size(); ++i) mine += (i ? "x" : "") + std::to_string(rad[i]);
        if (got >= 1 && fl == lanes_signed && (got == 1 || mine == rs)) cost *= 1e-3;
      }
      if (pair) pair_plans.emplace_back(cost, p);
      if (cost < best_cost) {
        best_cost = cost;
        best = p;
      }
    }
  }
  if (best_cost >= 1e30) return false;
  if (best.hx_pair != 0 && skip_pairs > 0) {
    // the caller's earlier choices did not compile within the pair's register budget (jit_spec_kernel): the next best plan with
    // other radices
    std::stable_sort(pair_plans.begin(), pair_plans.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    std::vector<std::vector<int>> seen;
    for (const auto& c : pair_plans) {
      if (std::find(seen.begin(), seen.end(), c.second.radices) != seen.end()) continue;
      if (static_cast<int>(seen.size()) == skip_pairs) {
        *out = c.second;
        return true;
      }
      seen.push_back(c.second.radices);
    }
    return false;
  }
  *out = best;
  return true;
}

bool choose_strided_params(int precision, long long n, long long inner_count, size_t max_lds, wg_params* out,
                           bool column_both, int want_fpw) {
  const int es = elem_bytes_of(precision);
  const bool f64 = precision == PFFT_PRECISION_F64;
  if (n < 2 || inner_count < 2) return false;
  wg_params p;
  p.precision = precision;
  p.n = static_cast<int>(n);
  p.radices = pick_radices(precision, n, (f64 && n > 32) ? 16 : 32);
  if (p.radices.empty()) return false;
  if (n <= 32) p.radices = {static_cast<int>(n)};  // short columns: one register pass, whatever the factorisation
  p.aux = 2;
  p.twm = 0;
  if (p.radices.size() == 1) {
    // one lane per FFT, a wave covers 64 adjacent columns (the reference's WORKITEM tier on strided data); no LDS
    p.fpw = 64;
    p.wg = 64;
    p.regs = p.n;
    p.occ = f64 ? (p.n <= 16 ? 2 : 1) : (p.n <= 16 ? 4 : 2);
    *out = p;
    return want_fpw == 0 || want_fpw == p.fpw;
  }
  const size_t lds_cap = std::min<size_t>(max_lds, 128 * 1024);
  const int e_cap = f64 ? 16 : 32;
  // planner experiments (tools/jit_sweep_strided.py): PFFT_JIT_STRIDED_FORCE=n:fpw:lanes_per_fft:r0xr1x...[:twl]
  const jit_knobs kn = jit_knobs::from_env();
  if (const char* e = kn.strided_force) {
    long long fn = 0;
    int ffpw = 0, ftpf = 0, ftwl = 0;
    char rad[64] = {0};
    const int got = std::sscanf(e, "%lld:%d:%d:%63[0-9x]:%d", &fn, &ffpw, &ftpf, rad, &ftwl);
    if (got >= 4 && fn == n && ffpw > 0 && ftpf > 0 && (want_fpw == 0 || want_fpw == ffpw)) {
      std::vector<int> rs;
      long long prod = 1;
      for (const char* c = rad; *c != '\0';) {
        rs.push_back(std::atoi(c));
        prod *= rs.back();
        while (*c != '\0' && *c != 'x') ++c;
        if (*c == 'x') ++c;
      }
      if (prod == n && rs.size() >= 2 && rs.size() <= static_cast<size_t>(MAX_PASSES)) {
        p.radices = rs;
        p.fpw = ffpw;
        p.wg = ftpf * ffpw;
        p.regs = 0;
        for (int r : rs) p.regs = std::max<int>(p.regs, static_cast<int>((n / r + ftpf - 1) / ftpf) * r);
        p.occ = f64 ? (p.regs <= 16 ? 2 : 1) : 2;
        p.twl = got >= 5 ? ftwl : 0;
        if (p.wg <= 1024 && spec_lds_bytes(p) <= max_lds) {
          *out = p;
          return true;
        }
      }
    }
  }
  // FPW adjacent FFTs: 256-byte HBM segments when LDS allows (32 fp32 columns; 16 fp64 columns only for stages that
  // are column-shaped on both sides -- a row-shaped fp64 side gets worse with more rows per wave), else 128-byte
  int fpw_first = f64 ? (column_both ? 16 : 8) : 32;
  if (kn.strided_fpw != 0) fpw_first = kn.strided_fpw;  // experiments (PFFT_JIT_STRIDED_FPW)
  if (kn.strided_lds_kib > 0) {  // experiments (PFFT_JIT_STRIDED_LDS_KIB): the widest group whose image stays below
    while (fpw_first > 4 && static_cast<size_t>(n) * fpw_first * es > (static_cast<size_t>(kn.strided_lds_kib) << 10)) fpw_first /= 2;
  }
  // (down to ONE FFT per work-group: a strided or batch-interleaved transform longer than half the LDS -- fp32 10 241 ...
  //  20 480 points, fp64 5121 ... 10 240 -- has no other single-kernel plan, the generic tier needs two images)
  for (int fpw = want_fpw > 0 ? want_fpw : fpw_first; fpw >= 1; fpw /= 2) {
    if (want_fpw > 0 && fpw != want_fpw) break;
    if (fpw / 2 >= inner_count && fpw > 2) continue;  // narrow stages: do not idle more than half of the group
    if (static_cast<size_t>(n) * fpw * es > (fpw == 1 ? max_lds : lds_cap)) continue;
    if (fpw == 1 && static_cast<size_t>(n) * 2 * es <= max_lds) continue;  // (the generic tier's length: unchanged)
    int wg_max = (f64 && fpw > 1) ? 512 : 1024;  // (one long fp64 transform per work-group: 10240 points on 1024 lanes)
    if (kn.strided_wg != 0) wg_max = kn.strided_wg;  // experiments (PFFT_JIT_STRIDED_WG)
    const int t_max = wg_max / fpw;
    int max_r = 0;
    for (int r : p.radices) max_r = std::max(max_r, r);
    const int target = std::max<int>(1, std::min<long long>(t_max, n / 16));
    const int reg_cap = std::max(max_r, e_cap + e_cap / 4);
    lane_choice lc = pick_lanes(p.radices, n, std::max(1, target / 2), t_max, target, 16, reg_cap, fpw, wg_max);
    if (lc.tpf == 0) continue;
    if (lc.tpf * fpw < 64) {  // at least one full wave per work-group: spread the FFT over more lanes
      lc.tpf = 64 / fpw;
      lc.regs = 0;
      for (int r : p.radices) lc.regs = std::max<int>(lc.regs, static_cast<int>((n / r + lc.tpf - 1) / lc.tpf) * r);
    }
    p.fpw = fpw;
    p.wg = lc.tpf * fpw;
    p.regs = lc.regs;
    p.occ = f64 ? (p.regs <= 16 ? 2 : 1) : 2;
    {  // leading twiddle tables in LDS (auto_twl_strided's rule)
      const size_t cu_lds = 160 * 1024, base = static_cast<size_t>(n) * fpw * es;
      const size_t waves = static_cast<size_t>((p.wg + 63) / 64), before = cu_lds / base;
      for (int k = static_cast<int>(p.radices.size()) - 1; k >= 1 && p.twl == 0; --k) {
        size_t extra = 0, ns = 1;
        for (int i = 0; i <= k; ++i) {
          if (i > 0) extra += ns * static_cast<size_t>(p.radices[static_cast<size_t>(i)] - 1);
          ns *= static_cast<size_t>(p.radices[static_cast<size_t>(i)]);
        }
        extra *= es;
        if (base + extra > std::min<size_t>(cu_lds, max_lds)) continue;
        const size_t after = cu_lds / (base + extra);
        if (extra <= 16 * 1024 && (after == before || after * waves >= 16)) p.twl = k;
      }
    }
    *out = p;
    return true;
  }
  return false;
}

size_t strided_hx_lds_bytes(const wg_params& p) {
  int h = 0;
  for (size_t i = 1; i < p.radices.size(); ++i) h = std::max(h, ((p.radices[i] + 1) / 2) * (p.n / p.radices[i]));
  size_t tw = 0, ns = 1;
  for (size_t i = 0; i < p.radices.size(); ++i) {
    if (i > 0 && static_cast<int>(i) <= p.twl) tw += ns * static_cast<size_t>(p.radices[i] - 1);
    ns *= static_cast<size_t>(p.radices[i]);
  }
  return (static_cast<size_t>(h) * static_cast<size_t>(p.fpw) + tw) * elem_bytes_of(p.precision);
}

bool choose_strided_wide_base(int precision, long long n, int fpw, wg_params* out) {
  if (n < 64 || n > 4096 || fpw < 2) return false;
  wg_params p;
  p.precision = precision;
  p.n = static_cast<int>(n);
  // (radices up to 16: a radix-32 butterfly holds 32 values beside its temporaries -- the budget of such a group is 32 values and
  //  ~62 registers of everything else)
  p.radices = pick_radices(precision, n, 16);
  if (p.radices.size() < 2 || p.radices.size() > static_cast<size_t>(MAX_PASSES)) return false;
  {
    // the planner's radices minimise passes and balance them; what decides here is the widest pass on 64 lanes per transform
    // (1024 fp32 / 512 fp64 lanes): fp32 2000 as 10.8.5.5 holds 40 values per lane, as 16.5.5.5 it holds 35.  When the planner's
    // set is beyond the 36 the candidates accept, the set with the narrowest widest pass (fewest passes, then largest first).
    auto widest = [&](const std::vector<int>& rs) {
      long long w = 0;
      for (int r : rs) w = std::max<long long>(w, ((n / r + 63) / 64) * r);
      return w;
    };
    if (widest(p.radices) > 36) {
      std::vector<int> best, cur;
      long long best_w = widest(p.radices);
      std::function<void(long long, int)> go = [&](long long rem, int max_r) {
        if (rem == 1) {
          if (cur.size() >= 2 && (widest(cur) < best_w || (widest(cur) == best_w && !best.empty() && cur.size() < best.size()))) {
            best = cur;
            best_w = widest(cur);
          }
          return;
        }
        if (cur.size() >= static_cast<size_t>(MAX_PASSES)) return;
        for (int r = std::min<long long>(max_r, rem); r >= 2; --r) {
          if (rem % r != 0) continue;
          cur.push_back(r);
          go(rem / r, r);
          cur.pop_back();
        }
      };
      go(n, 16);
      if (!best.empty()) p.radices = best;
    }
  }
  p.aux = 2;
  p.twm = 0;
  p.fpw = fpw;
  p.occ = 1;
  p.wg = 0;
  p.regs = 0;
  *out = p;
  return true;
}

std::vector<wg_params> strided_hx_candidates(const wg_params& base, size_t max_lds, bool wide, size_t min_bytes) {
  std::vector<wg_params> out;
  const jit_knobs kn = jit_knobs::from_env();
  const int es = elem_bytes_of(base.precision);
  const bool f64 = base.precision == PFFT_PRECISION_F64;
  const size_t full = static_cast<size_t>(base.n) * static_cast<size_t>(base.fpw) * es;
  if (!kn.strided_hx || base.radices.size() < 2 || base.fpw < 2 || base.staged != 0 ||
      (min_bytes != 0 ? full < min_bytes : full <= (static_cast<size_t>(kn.strided_hx_min_kib) << 10))) {
    return out;
  }
  // wide: a group beyond the LDS (choose_strided_wide_base) -- ONE work-group per CU on a half image, or nothing
  if (wide && (!kn.strided_hx_wide || full <= std::min<size_t>(max_lds, 128 * 1024))) return out;
  const int per_cu_lo = wide ? 1 : 2, per_cu_hi = wide ? 1 : 4;
  const size_t cu_lds = std::min<size_t>(max_lds, 160 * 1024);
  const size_t tables = f64 ? 8 * 1024 : 4 * 1024;  // store-modifier tables + allocation granularity, per work-group
  // the order of the radices: the planner's own (a small last radix keeps the store modifier cheap) while half an image of
  // it lets two work-groups share the CU; else the order with the smallest image -- pass 0 is the one pass whose inputs do
  // not come through the image, and the image is ceil(R / 2) / R of the group for the others
  auto image_of = [&](const std::vector<int>& o) {
    size_t h = 0;
    for (size_t i = 1; i < o.size(); ++i) {
      h = std::max<size_t>(h, static_cast<size_t>((o[i] + 1) / 2) * static_cast<size_t>(base.n / o[i]));
    }
    return h * static_cast<size_t>(base.fpw) * es;
  };
  std::vector<int> rad = base.radices;
  if (image_of(rad) + tables > (wide ? cu_lds : cu_lds / 2)) {
    std::vector<int> order = base.radices;
    std::sort(order.begin(), order.end());
    size_t best_h = image_of(rad);
    do {
      if (image_of(order) < best_h) {
        best_h = image_of(order);
        rad = order;
      }
    } while (std::next_permutation(order.begin(), order.end()));
  }
  int rmax = 0;
  for (int r : rad) rmax = std::max(rmax, r);
  std::vector<std::pair<double, wg_params>> ranked;
  for (int tpf = std::max(1, 64 / base.fpw); tpf * base.fpw <= 1024; ++tpf) {
    int regs = 0;
    double slots = 0.0;
    for (int r : rad) {
      const long long nb = base.n / r;
      const long long bpt = (nb + tpf - 1) / tpf;
      regs = std::max<int>(regs, static_cast<int>(bpt) * r);
      slots += static_cast<double>(tpf) * static_cast<double>(bpt) * r / static_cast<double>(base.n);
    }
    const int wg = tpf * base.fpw, waves = (wg + 63) / 64;
    for (int per_cu = per_cu_lo; per_cu <= per_cu_hi; ++per_cu) {
      wg_params p = base;
      p.radices = rad;
      p.wg = wg;
      p.regs = regs;
      p.twl = 0;
      p.hx_strided = per_cu;
      const size_t budget_lds = cu_lds / static_cast<size_t>(per_cu);
      if (strided_hx_lds_bytes(p) + tables > budget_lds) continue;
      for (int k = static_cast<int>(rad.size()) - 1; k >= 1; --k) {  // leading twiddle tables in LDS while they fit
        wg_params q = p;
        q.twl = k;
        if (strided_hx_lds_bytes(q) - strided_hx_lds_bytes(p) <= 16 * 1024 && strided_hx_lds_bytes(q) + tables <= budget_lds) {
          p.twl = k;
          break;
        }
      }
      const int wps = (per_cu * waves + 3) / 4;
      if (wps > 8) continue;
      // fp32: only plans that keep 14 or more waves on the CU.  Measured (tools/perf_stage_hx.py, profiles/r6_stage_hx_first.txt;
      // fraction of the HBM peak, LDS-resident -> register-resident): 768 x 16 on 2 x 512 lanes 0.453 -> 0.541, 660 x 16 on
      // 2 x 480 lanes 0.513 -> 0.551, 68640 = 104 x 660 0.289 -> 0.301 -- but 800 x 16 on 2 x 320 lanes 0.438 -> 0.438 and 1728 x 8
      // on 2 x 384 lanes 0.233 -> 0.215: ten or twelve waves of 160-register kernels hide less than the thirteen of the
      // LDS-resident one.  (fp64 has no such pattern: 660 x 8 on 2 x 176 lanes 0.469 -> 0.527, 768 x 8 on 2 x 256 lanes a tie.)
      if (!f64 && per_cu * waves < 14 && kn.strided_hx_force == nullptr) continue;
      const int budget = std::min(256, (512 / wps) / 8 * 8);
      // the group's values, one butterfly's temporaries, the store modifier's powers, addresses (the compiler has the last
      // word: a kernel that needs scratch at this occupancy is dropped for the next candidate, jit_strided_kernel)
      // Measured with hipcc -Rpass-analysis (tools/jit_strided_hx_dump.cpp + tools/kres.py): beside the values a kernel of
      // this shape needs 60-70 registers in fp32 and ~100 in fp64 -- fp32 11.10.6 x 16 on 480 lanes (30 values) 121 VGPRs,
      // 10.10.10 x 16 on 560 lanes (30 values) 131 unconstrained and 26 spilled at 96, on 320 lanes (50 values) 205
      // (16.8.8 x 16 on 512 lanes, 32 values: 115 without the store modifier, 16 spilled with it -- the compile check decides)
      const int need = regs * (f64 ? 4 : 2) + (f64 ? 100 : 62) + std::max(0, rmax - 16) * (f64 ? 4 : 2);
      bool forced = false;
      if (const char* e = kn.strided_hx_force) {  // experiments: "tpf:per_cu", whatever the register estimate says
        int ft = 0, fk = 0;
        forced = std::sscanf(e, "%d:%d", &ft, &fk) == 2 && ft == tpf && fk == per_cu;
      }
      // (a wide group has no LDS-resident plan to fall back on and may spill a few registers, jit.cpp: up to 36 fp32 values per
      //  lane are handed to the compiler -- PFFT_JIT_STRIDED_HX_WIDE_SLACK registers beyond the estimate)
      if (need > budget + (wide ? static_cast<int>(kn.strided_hx_wide_slack) : 0) && !forced) continue;
      p.occ = wps;
      const double idle = slots / static_cast<double>(rad.size()) - 1.0;
      const double lane_waste = static_cast<double>(waves * 64) / static_cast<double>(wg) - 1.0;
      double cost = (1.0 + idle) * (1.0 + lane_waste);
      // 16 waves per CU in two or more work-groups is where the packed pairs and the registered stage pairs sit
      cost *= 1.0 + 0.04 * std::abs(per_cu * waves - 16);
      cost *= 1.0 + 0.3 * std::max(0.0, static_cast<double>(need) / budget - 0.85);  // at the edge of the budget
      if (forced) cost *= 1e-3;
      ranked.emplace_back(cost, p);
    }
  }
  std::stable_sort(ranked.begin(), ranked.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  for (const auto& c : ranked) {
    bool dup = false;
    // (one plan per register load: 544, 560 and 576 lanes holding 30 values each are the same kernel to the allocator)
    for (const wg_params& o : out) dup = dup || (o.regs == c.second.regs && o.hx_strided == c.second.hx_strided);
    if (!dup) out.push_back(c.second);
    if (out.size() >= 3) break;
  }
  return out;
}

bool choose_rows2d_params(int precision, long long n1, long long n0, size_t max_lds, wg_params* out, int rc_mask) {
  const int es = elem_bytes_of(precision);
  const int emax = precision == PFFT_PRECISION_F64 ? 16 : 32;  // complex elements a lane may hold
  if (n1 < 32 || n1 > 16384 || n0 < 4) return false;
  // planner experiments (tools/jit_sweep_strided.py): PFFT_JIT_ROWS2D_FORCE=n1:rc:lanes:r0x...xr_last[:twl]
  if (const char* e = jit_knobs::from_env().rows2d_force) {
    long long fn = 0;
    int frc = 0, fwg = 0, ftwl = 0;
    char rad[64] = {0};
    const int got = std::sscanf(e, "%lld:%d:%d:%63[0-9x]:%d", &fn, &frc, &fwg, rad, &ftwl);
    if (got >= 4 && fn == n1 && frc > 0 && fwg > 0 && n0 % frc == 0 && (rc_mask & frc) != 0) {
      wg_params p;
      long long prod = 1;
      for (const char* c = rad; *c != '\0';) {
        p.radices.push_back(std::atoi(c));
        prod *= p.radices.back();
        while (*c != '\0' && *c != 'x') ++c;
        if (*c == 'x') ++c;
      }
      const long long nbl = p.radices.empty() ? 0 : n1 / p.radices.back();
      if (prod == n1 && p.radices.size() >= 2 && nbl % fwg == 0 && fwg % frc == 0) {
        p.precision = precision;
        p.n = static_cast<int>(n1);
        p.wg = fwg;
        p.fpw = frc;
        if (const int r0 = p.radices[0]; r0 >= 8 && (r0 & (r0 - 1)) == 0) {
          p.pads = 16;
          p.padw = 1;
        }
        p.twm = 0;
        p.aux = 2;
        p.staged = 0;
        const long long tpf = fwg / frc;
        p.regs = static_cast<int>(nbl / fwg) * frc * p.radices.back();
        for (size_t i = 0; i + 1 < p.radices.size(); ++i) {
          const long long nb = n1 / p.radices[i];
          p.regs = std::max<int>(p.regs, static_cast<int>((nb + tpf - 1) / tpf) * p.radices[i]);
        }
        p.occ = p.regs * (es / 4) > 96 ? 2 : (p.regs * (es / 4) > 48 ? 3 : 4);
        p.twl = got >= 5 ? ftwl : 0;
        if (spec_lds_bytes(p) <= max_lds && p.regs <= 2 * emax) {
          *out = p;
          return true;
        }
      }
    }
  }
  double best_score = 1e30;
  wg_params best;
  for (int rc : {8, 4, 2}) {
    if (n0 % rc != 0 || n0 / rc < 2 || (rc_mask & rc) == 0) continue;
    for (int rl = 2; rl <= 16; ++rl) {
      if (n1 % rl != 0) continue;
      const long long nbl = n1 / rl;
      std::vector<int> rest = pick_radices(precision, nbl, 16);
      if (rest.empty() || (rest.size() == 1 && rest[0] == 1)) continue;
      if (static_cast<int>(rest.size()) + 1 > MAX_PASSES) continue;
      for (int bptl : {1, 2, 4}) {
        if (nbl % bptl != 0) continue;
        const long long wg = nbl / bptl;
        if (wg < 64 || wg > 1024 || wg % rc != 0) continue;
        const int e_last = rc * rl * bptl;
        if (e_last > emax) continue;
        const long long tpf = wg / rc;
        int regs = e_last;
        double slots = 0.0;
        for (int r : rest) {
          const long long nb = n1 / r;
          const long long bpt = (nb + tpf - 1) / tpf;
          regs = std::max<int>(regs, static_cast<int>(bpt) * r);
          slots += static_cast<double>(tpf * bpt * r) / static_cast<double>(n1);
        }
        if (regs > emax) continue;
        wg_params p;
        p.precision = precision;
        p.n = static_cast<int>(n1);
        p.radices = rest;
        p.radices.push_back(rl);
        p.wg = static_cast<int>(wg);
        p.fpw = rc;
        // same padding rule as the packed planner: +1 per 16 behind a power-of-two first radix >= 8
        if (const int r0 = p.radices[0]; r0 >= 8 && (r0 & (r0 - 1)) == 0) {
          p.pads = 16;
          p.padw = 1;
        }
        p.twm = 0;
        p.occ = regs * (es / 4) > 96 ? 2 : (regs * (es / 4) > 48 ? 3 : 4);
        p.aux = 2;
        p.staged = 0;
        p.regs = regs;
        // twiddle tables of the leading passes in LDS while they stay below 16 KiB
        p.twl = 0;
        size_t tw = 0, ns = 1;
        for (size_t i = 0; i < p.radices.size(); ++i) {
          if (i > 0) {
            tw += ns * static_cast<size_t>(p.radices[i] - 1) * es;
            if (tw <= 16 * 1024) p.twl = static_cast<int>(i);
          }
          ns *= static_cast<size_t>(p.radices[i]);
        }
        // ... but not at the price of the second work-group per CU (fp64 1024: 86 KiB with two tables 5.0 TB/s, 73 KiB
        // with one 6.2)
        while (p.twl > 0 && spec_lds_bytes(p) > 80 * 1024) --p.twl;
        const size_t lds = spec_lds_bytes(p);
        if (lds > max_lds) continue;
        // wider column radix = wider segments in pass 2, as long as two work-groups still fit a CU; then few idle
        // lane slots, whole waves, and ~256 lanes
        // fewest LDS exchanges first (every extra pass moves RC rows through LDS once more), then few idle lane
        // slots in ragged passes, then the wider column radix (wider segments in pass 2) while two work-groups
        // still fit a CU, whole waves, ~256 lanes
        double score = 1.0 * static_cast<double>(p.radices.size());
        score += 0.5 * (slots / static_cast<double>(rest.size()) - 1.0);
        score += (rc == 8 ? 0.0 : rc == 4 ? 0.15 : 0.4) + (lds > 80 * 1024 ? 0.5 : 0.0);
        score += (wg % 64 != 0 ? 0.1 : 0.0) + 0.0003 * std::abs(static_cast<double>(wg) - 256.0);
        if (score < best_score) {
          best_score = score;
          best = p;
        }
      }
    }
  }
  if (best_score >= 1e30) return false;
  *out = best;
  return true;
}

std::string wg_cfg_type_name(const wg_params& p) {
  std::ostringstream s;
  s << "pfa::wg_cfg<" << (p.precision == PFFT_PRECISION_F64 ? "double" : "float") << ", pfa::radix_list<";
  for (size_t i = 0; i < p.radices.size(); ++i) s << (i ? ", " : "") << p.radices[i];
  s << ">, " << p.wg << ", " << p.fpw << ", " << p.pads << ", " << p.padw << ", " << p.twm << ", " << p.occ << ", "
    << p.aux << ", " << p.staged << ", " << p.twl << ">";
  return s.str();
}

bool choose_nd_params(int precision, const std::vector<long long>& dims, size_t max_lds, nd_kernel* out) {
  const int es = elem_bytes_of(precision);
  const bool f64 = precision == PFFT_PRECISION_F64;
  if (dims.size() < 2) return false;
  long long ntot = 1;
  for (long long d : dims) {
    if (d < 1) return false;
    ntot *= d;
    // up to one work-group per CU: still ahead of separate passes (128x128 fp32 2.9 -> 3.6 TB/s, fp64 64x128 2.8 -> 4.1)
    if (ntot * es > 128 * 1024) return false;
  }
  if (ntot < 4) return false;
  nd_kernel p;
  std::vector<int> all;
  for (long long d : dims) {
    p.dims.push_back(static_cast<int>(d));
    std::vector<int> r;
    if (d > 1) {
      r = pick_radices(precision, d, 16);
      if (r.empty()) return false;
    }
    all.insert(all.end(), r.begin(), r.end());
    p.radices.push_back(r);
  }
  if (all.empty()) return false;
  const int e_target = 16;
  const int target = std::max<int>(1, static_cast<int>(ntot / e_target));
  int max_r = 0;
  bool pow2 = true;
  for (int r : all) {
    max_r = std::max(max_r, r);
    pow2 = pow2 && (r & (r - 1)) == 0;
  }
  const int occ_lanes = std::min<int>(64, static_cast<int>((ntot * es + 159) / 160));
  const int reg_cap = std::max(max_r, e_target + e_target / 2);
  const lane_choice lc = pick_lanes(all, ntot, std::max({1, target / 2, occ_lanes}),
                                    std::min<int>(1024, std::max(2 * target, occ_lanes)), target, e_target, reg_cap, 0,
                                    1024);
  if (lc.tpf == 0) return false;
  p.regs = lc.regs;
  if (pow2) {
    p.pads = 16;
    p.padw = 1;
  }
  spec_kernel& k = p.k;
  k.precision = precision;
  k.n = static_cast<int>(ntot);
  k.fpw = std::max(1, 256 / lc.tpf);
  auto lds_bytes = [&]() {
    const long long last = ntot - 1;
    const long long per = (p.pads == 0 ? last : last + ((last / p.pads) * p.padw)) + 1 + (p.pads == 0 ? 0 : p.padw);
    return static_cast<size_t>(per) * static_cast<size_t>(k.fpw) * es;
  };
  while (k.fpw > 1 && lds_bytes() > std::min<size_t>(max_lds, 40 * 1024)) --k.fpw;
  if (64 % lc.tpf == 0) {
    const int per_wave = 64 / lc.tpf;
    k.fpw = std::max(per_wave, k.fpw / per_wave * per_wave);
  }
  if (lds_bytes() > max_lds) return false;
  k.wg = lc.tpf * k.fpw;
  k.lds_bytes = lds_bytes();
  k.groups_per_wg = 1;
  k.n_radices = 0;
  int tw = 0;
  for (const std::vector<int>& r : p.radices) {
    int ns = 1;
    for (size_t i = 0; i < r.size(); ++i) {
      if (i > 0) tw += ns * (r[i] - 1);
      ns *= r[i];
    }
  }
  k.tw_total = tw;
  p.occ = f64 ? (p.regs <= 16 ? 2 : 1) : (p.regs <= 16 ? 4 : (p.regs <= 24 ? 3 : 2));
  *out = p;
  return true;
}

std::string nd_cfg_type_name(const nd_kernel& p) {
  std::ostringstream s;
  s << "pfa::nd_cfg<" << (p.k.precision == PFFT_PRECISION_F64 ? "double" : "float") << ", " << p.k.n << ", " << p.k.wg
    << ", " << p.k.fpw << ", " << p.pads << ", " << p.padw << ", " << p.occ << ", 2";
  // passes: last (contiguous) dimension first, like the reference's dimension loop; twiddle tables in the same order
  int tw_base = 0;
  for (int d = static_cast<int>(p.dims.size()) - 1; d >= 0; --d) {
    long long stride = 1;
    for (size_t e = static_cast<size_t>(d) + 1; e < p.dims.size(); ++e) stride *= p.dims[e];
    const std::vector<int>& r = p.radices[static_cast<size_t>(d)];
    int ns = 1, off = 0;
    for (size_t i = 0; i < r.size(); ++i) {
      s << ", pfa::nd_pass<" << r[i] << ", " << ns << ", " << p.dims[static_cast<size_t>(d)] << ", " << stride << ", "
        << tw_base + off << ">";
      if (i > 0) off += ns * (r[i] - 1);
      ns *= r[i];
    }
    // off now holds the entries of passes 1..last-1; add the last pass's share
    int total = 0, m = 1;
    for (size_t i = 0; i < r.size(); ++i) {
      if (i > 0) total += m * (r[i] - 1);
      m *= r[i];
    }
    tw_base += total;
  }
  s << ">";
  return s.str();
}

}  // namespace pfa
