// plan_t: the 1-D planner -- packed / UNPACKED / strided work-group stages, and the GLOBAL tier: four-step pairs, the
// three-stage plan, the XCD-local single launch, measured planning (plan.hpp).
#include "plan.hpp"

namespace pfa {

/// Width of the intermediate's tiles -- i.e. the group width its stage A must have -- when `fb` is the four-step
/// stage B of length n2: its own group width (square tiles, launch_tin) or, `wide`, twice that (launch_tin_w).
/// 0 when the entry has no such form or the length does not divide into those tiles.
int plan_t::pair_tile(const strided_kernel* fb, long long n2, bool wide) {
  const int t = wide ? (fb->launch_tin_w != nullptr ? fb->tin_w : 0) : (fb->launch_tin != nullptr ? fb->fpw : 0);
  if (t <= 0 || (t & (t - 1)) != 0 || n2 % t != 0 || (n2 / fb->radices[0]) % t != 0) return 0;
  return t;
}

/// Measured planning of the four-step split (PFFT_PLAN_MEASURE=1): every n1 x n2 with both factors in 32 ... 4096, no
/// more than 16 : 1 apart, that the strided tier can run -- the planner's own choice first -- is committed as a plan of its own (this descriptor, a batch
/// of 256 MiB, plan_t's forced_n1), timed forward on the plan's stream, and the winner is recorded next to the code
/// objects like the radix choices (`choice_split_<arch>_<f32|f64>_<n>.txt` holds "n1 n2").
long long plan_t::measured_split(long long n, long long count, long long static_n1) {
  const std::string arch = jit_device_arch();
  const std::vector<int> rec = plan_choice_lookup(arch, desc.precision, n, 1 << 20, true);
  if (rec.size() == 2 && strided_fpw(rec[0], rec[1]) > 0 && strided_fpw(rec[1], rec[0]) > 0) return rec[0];
  std::vector<long long> cands{static_n1};
  for (long long c = 32; c <= 4096; ++c) {
    if (n % c != 0 || c == static_n1) continue;
    const long long m = n / c;
    if (m < 32 || m > 4096 || std::max(c, m) > 16 * std::min(c, m) || strided_fpw(c, m) <= 0 || strided_fpw(m, c) <= 0) continue;
    cands.push_back(c);
  }
  if (cands.size() == 1) return static_n1;
  const size_t eb = elem_bytes();
  const size_t per = static_cast<size_t>(n) * eb;
  const long long batch = std::max<long long>(1, std::min<long long>(count, static_cast<long long>((size_t{256} << 20) / per)));
  const size_t bytes = static_cast<size_t>(batch) * per;
  measure_scratch ms_;  // (freed on every way out, a throwing hip_check included)
  if (!ms_.alloc(bytes)) return static_n1;  // no room to measure: the static rule
  void *const in = ms_.in, *const out = ms_.out;
  fill_uniform(in, bytes);
  pfft_desc_t d = desc;
  d.number_of_transforms = static_cast<uint64_t>(batch);
  d.placement = PFFT_OUT_OF_PLACE;
  d.forward_offset = 0;
  d.backward_offset = 0;
  const size_t half = bytes / 2;  // (split storage: the two planes inside the same allocations)
  const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
  hip_check(hipEventCreate(&ms_.e0), "hipEventCreate");
  hip_check(hipEventCreate(&ms_.e1), "hipEventCreate");
  const hipEvent_t e0 = ms_.e0, e1 = ms_.e1;
  long long best = static_n1;
  double best_ms = 1e30;
  for (long long c : cands) {
    float ms = 0.f;
    bool ok = true;
    try {
      plan_t sub(d, stream, c);
      for (int rep = 0; rep < 7 && ok; ++rep) {
        if (rep == 2) ok = hipEventRecord(e0, stream) == hipSuccess;
        sub.execute(PFFT_FORWARD, in, split ? static_cast<char*>(in) + half : nullptr, out,
                    split ? static_cast<char*>(out) + half : nullptr);
      }
      ok = ok && hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
           hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    } catch (const std::exception&) {
      ok = false;
      (void)hipStreamSynchronize(stream);
    }
    if (kn.jit_verbose) {
      std::fprintf(stderr, "[portfft_amd plan] n=%lld split %lld x %lld %.3f ms per %lld transforms%s\n", n, c, n / c, ms / 5,
                   batch, ok ? "" : " (failed)");
    }
    if (ok && ms < best_ms) {
      best_ms = ms;
      best = c;
    }
  }
  if (best_ms < 1e30) plan_choice_store(arch, desc.precision, n, {static_cast<int>(best), static_cast<int>(n / best)}, true);
  return best;
}

/// uniform(-1, 1) scalars: a 1 MiB host block replicated by doubling copies on the plan's stream
void plan_t::fill_uniform(void* dst, size_t bytes) {
  const size_t block = std::min<size_t>(bytes, size_t{1} << 20);
  std::vector<unsigned char> h(block);
  unsigned long long z = 0x9E3779B97F4A7C15ull;
  const size_t scalars = block / static_cast<size_t>(scalar_bytes());
  for (size_t i = 0; i < scalars; ++i) {
    z = z * 6364136223846793005ull + 1442695040888963407ull;
    const double v = static_cast<double>(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    if (scalar_bytes() == 8) {
      reinterpret_cast<double*>(h.data())[i] = v;
    } else {
      reinterpret_cast<float*>(h.data())[i] = static_cast<float>(v);
    }
  }
  hip_check(hipMemcpyAsync(dst, h.data(), block, hipMemcpyHostToDevice, stream), "hipMemcpy");
  hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
  for (size_t have = block; have < bytes; have *= 2) {
    hip_check(hipMemcpyAsync(static_cast<char*>(dst) + have, dst, std::min(have, bytes - have), hipMemcpyDeviceToDevice, stream),
              "hipMemcpy");
  }
}

/// Measured planning (PFFT_PLAN_MEASURE=1; the reference's rule is static, committed_descriptor_impl.hpp:210-313): the
/// radix sequence of a runtime-specialised packed length is the fastest of the planner's top candidates
/// (jit.cpp: spec_radix_candidates), timed here on the plan's stream over 256 MiB of random data, and recorded next to
/// the code objects in the JIT cache -- later commits (this process or another) read the record instead of measuring.
std::vector<int> plan_t::measured_radices(long long n) {
  const std::string arch = jit_device_arch();
  std::vector<int> choice = plan_choice_lookup(arch, desc.precision, n);
  if (!choice.empty()) return choice;
  std::vector<std::vector<int>> cands = spec_radix_candidates(desc.precision, n, max_lds);
  if (cands.empty()) return choice;
  // A prime factor P of 17 ... 61 makes the lanes per transform a candidate too (a sequence ends in "0, lanes"): the
  // prime's pass has n / P butterflies, and whether a transform should take that many lanes, twice or half as many,
  // with the prime first or last, is not something a rule gets right below 37 (tools/probes/prime_rule2.sh: 31 x 32
  // 0.53 -> 0.67 but 31 x 31 in fp64 0.53 -> 0.36 with the rule of the primes above)
  {
    int big = 0;
    for (int r : cands[0]) big = std::max(big, r);
    bool prime = big >= 17;
    for (int q = 2; q * q <= big; ++q) prime = prime && big % q != 0;
    const long long nb = n / std::max(big, 1);
    if (prime && nb >= 8 && nb <= 128) {
      int t0 = 16;
      while (t0 < nb && t0 < 128) t0 *= 2;
      std::vector<std::vector<int>> seqs(cands.begin(), cands.begin() + std::min<size_t>(cands.size(), 3));
      if (nb <= 32) {
        seqs.push_back({static_cast<int>(nb), big});
        seqs.push_back({big, static_cast<int>(nb)});
      }
      for (const std::vector<int>& q : seqs) {
        for (int t : {t0 / 2, t0, 2 * t0}) {
          if (t < 16 || t > 256) continue;
          std::vector<int> v = q;
          v.push_back(0);
          v.push_back(t);
          cands.push_back(v);
        }
      }
    }
  }
  if (cands.size() == 1) {
    plan_choice_store(arch, desc.precision, n, cands[0]);
    return cands[0];
  }
  const size_t eb = elem_bytes();
  const long long batch = std::max<long long>(1, static_cast<long long>((size_t{256} << 20) / (static_cast<size_t>(n) * eb)));
  const size_t bytes = static_cast<size_t>(batch) * static_cast<size_t>(n) * eb;
  measure_scratch ms_;
  if (!ms_.alloc(bytes)) return choice;  // no room to measure: the static rule
  void *const in = ms_.in, *const out = ms_.out;
  fill_uniform(in, bytes);
  hip_check(hipEventCreate(&ms_.e0), "hipEventCreate");
  hip_check(hipEventCreate(&ms_.e1), "hipEventCreate");
  const hipEvent_t e0 = ms_.e0, e1 = ms_.e1;
  double best_ms = 1e30;
  for (const std::vector<int>& r : cands) {
    std::string why;
    const spec_kernel* k = jit_spec_kernel(desc.precision, n, false, max_lds, &why, false, &r);
    if (k == nullptr) continue;
    const std::vector<int> radices_only(k->radices, k->radices + k->n_radices);
    struct device_mem {  // (freed on every way out, a throwing hip_check included)
      void* p = nullptr;
      ~device_mem() {
        if (p != nullptr) (void)hipFree(p);
      }
    } tw_mem;
    void*& tw = tw_mem.p;
    {
      std::vector<char> host;
      if (desc.precision == PFFT_PRECISION_F64) {
        const auto t = host_twiddles<double>(radices_only);
        host.assign(reinterpret_cast<const char*>(t.data()), reinterpret_cast<const char*>(t.data() + t.size()));
      } else {
        const auto t = host_twiddles<float>(radices_only);
        host.assign(reinterpret_cast<const char*>(t.data()), reinterpret_cast<const char*>(t.data() + t.size()));
      }
      if (hipMalloc(&tw, host.size()) != hipSuccess) continue;
      hip_check(hipMemcpy(tw, host.data(), host.size(), hipMemcpyHostToDevice), "hipMemcpy(twiddles)");
    }
    const long long groups = (batch + k->fpw - 1) / k->fpw;
    const unsigned grid = persistent_grid(nullptr, k->mfn[0], k->wg, k->lds_bytes, groups, k->groups_per_wg);
    float ms = 0.f;
    bool ok = true;
    for (int rep = 0; rep < 11 && ok; ++rep) {
      if (rep == 3) ok = hipEventRecord(e0, stream) == hipSuccess;
      ok = ok && jit_launch_spec(k, stream, grid, in, out, tw, batch, 1.0, 0) == hipSuccess;
    }
    ok = ok && hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
         hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    if (kn.jit_verbose) {
      std::string rs;
      for (int x : r) rs += std::to_string(x) + ".";
      std::fprintf(stderr, "[portfft_amd plan] n=%lld radices %s %.3f ms per %lld transforms%s\n", n, rs.c_str(), ms / 8,
                   batch, ok ? "" : " (failed)");
    }
    if (ok && ms < best_ms) {
      best_ms = ms;
      choice = r;
    }
  }
  if (!choice.empty()) plan_choice_store(arch, desc.precision, n, choice);
  return choice;
}

/// BATCH_INTERLEAVED on both sides (element i of transform b at i * B + b), length n = n1 * n2, B transforms --
/// `outer` such arrays n * B elements apart (the long column dimension of an N-D array: outer matrices, B adjacent
/// columns; outer = 1 for a batch-interleaved 1-D descriptor).
/// Stage A: for every (c, b): FFT over r of x[(r*n2 + c)*B + b], times W_n^{k1*c}, into scratch (same layout).
/// Stage B: for every (k1, b): FFT over c of scratch[(k1*n2 + c)*B + b] -> out[(k2*n1 + k1)*B + b].
/// Only taken when a single work-group would hold fewer than 16 (fp32) / 8 (fp64) columns of the whole length.
bool plan_t::plan_batch_interleaved_two_stage(std::vector<stage>& out, long long n, long long B, long long outer,
                                              int in_buf, int out_buf, const addressing& ia, const addressing& oa,
                                              double scale, int backward, pfft_dim_info_t* info) {
  const int full_fpw = desc.precision == PFFT_PRECISION_F64 ? 8 : 16;
  if (strided_fpw(n, B) >= full_fpw) return false;
  // An array of 4 GiB or more: the byte offsets of a stage's butterfly legs no longer fit the 32-bit scalar offset of the
  // buffer instructions.  Round 6: both stages then run the BIG forms of their kernels (strided_io_big: 64-bit leg offsets,
  // plain global accesses; compiled at commit whatever the length) -- fp32 N = 4096 x 131 136 fell to the generic tier before,
  // 0.08 of the HBM peak against 0.33 just below 4 GiB.  What stays 32 bits is a LANE's offset, the span of the first / last
  // pass's butterflies = 1 / radix of the array: up to 16 GiB with the radices (>= 8) of the stage lengths used here.
  const unsigned long long array_bytes = static_cast<unsigned long long>(n) * static_cast<unsigned long long>(B) * elem_bytes();
  const bool big = array_bytes >= 0xFFFFFFF0ull;
  if (big && (outer != 1 || !jit_enabled() || kn.no_big_bi || array_bytes > (16ull << 30) ||
              static_cast<unsigned long long>(n) * static_cast<unsigned long long>(B) >= (1ull << 32))) {
    return false;
  }
  const size_t need = static_cast<size_t>(n) * static_cast<size_t>(B) * static_cast<size_t>(outer) * elem_bytes();
  if (outer > 1 && need > global_chunk_bytes()) return false;  // the intermediate is as large as the data
  long long n1 = 0;
  for (long long c = static_cast<long long>(std::sqrt(static_cast<double>(n))); c >= 2; --c) {
    if (n % c != 0) continue;
    if (strided_fpw(c, (n / c) * B) >= full_fpw && strided_fpw(n / c, B) >= full_fpw) {
      n1 = c;
      break;
    }
  }
  if (n1 == 0) return false;
  // Round 6 (PFFT_BI_N1 sweeps, profiles/r6_bi_two_stage_split.txt, fraction of the HBM peak): the balanced split makes stage A of
  // N = 2048 a 32-point single-pass kernel that carries the store modifier on a radix-32 butterfly -- 32 x 64 0.219, 64 x 32 0.306,
  // **128 x 16 0.328**; N = 4096: 64 x 64 0.344, 128 x 32 0.352; 8192 / 16384: 128 x 64 / 128 x 128 are the best or tie it; fp64 2048
  // 0.286 -> 0.318, 4096 0.332 -> 0.353 (32 x 128: 0.368).  A 128-point stage A (8.16 x 64 columns, modifier tables in LDS) in front
  // of whatever is left: the rule for every length it divides.
  if (n % 128 == 0 && n / 128 >= 2 && strided_fpw(128, (n / 128) * B) >= full_fpw && strided_fpw(n / 128, B) >= full_fpw &&
      !kn.no_bi_n1_rule) {
    n1 = 128;
  }
  if (kn.bi_n1 > 0 && n % kn.bi_n1 == 0 && strided_fpw(kn.bi_n1, (n / kn.bi_n1) * B) >= full_fpw &&
      strided_fpw(n / kn.bi_n1, B) >= full_fpw) {
    n1 = kn.bi_n1;  // experiments (PFFT_BI_N1): the first factor of the two-stage BI plan
  }
  const long long n2 = n / n1;
  // the intermediate is written once and read once: keep it in the Infinity Cache when all of it fits
  // (measured with random data, tools/perf_cache.py: +4...13 % from 128 MiB of intermediate up; below that the
  //  streamed kernels are faster -- everything sits in the cache anyway -- so small problems keep them)
  const bool cached = cache_chunk_bytes() > 0 && need <= cache_chunk_bytes() && need >= cache_chunk_bytes() / 2;
  const strided_kernel* ka = nullptr;
  const strided_kernel* kb = nullptr;
  // (a batch count that is no multiple of a line -- 16 fp32 / 8 fp64 transforms: every row pitch of both stages is unaligned,
  //  policy 3 of aux_of_policy)
  const bool unal = !kn.no_unaligned_policy && (static_cast<unsigned long long>(B) * elem_bytes()) % 128 != 0 &&
                    array_bytes * static_cast<unsigned long long>(outer) >= (64ull << 20);
  if (big) {
    std::string why;
    ka = jit_strided_kernel(desc.precision, n1, n2 * B, true, 0, max_lds, &why, true, unal ? 3 : 0, 0, true);
    kb = jit_strided_kernel(desc.precision, n2, B, false, 0, max_lds, &why, true, unal ? 3 : 0, 0, true);
    if (ka == nullptr || kb == nullptr) jit_note("strided (big)", ka == nullptr ? n1 : n2, why);
  } else {
    ka = get_strided(n1, n2 * B, true, false, true, false, unal ? 3 : (cached ? 1 : 0));  // column-shaped on
    kb = get_strided(n2, B, false, false, true, false, unal ? 3 : (cached ? 2 : 0));      // both sides: wide
  }
  addressing a_in{ia.offset, n2 * B, 1, n * B};
  addressing a_out{0, n2 * B, 1, n * B};
  addressing b_in{0, B, 1, n2 * B};
  addressing b_out{oa.offset, n1 * B, 1, B};
  if (big) {
    // (strided_fits checks the 32-bit byte range of a whole group: what the BIG forms lift.  A lane's offset -- the span
    //  of one pass's butterflies -- must still fit: first and last radix of either kernel)
    auto lane_span_ok = [&](const strided_kernel* k) {
      if (k == nullptr || k->n_radices < 1) return false;
      const int r = std::min(k->radices[0], k->radices[k->n_radices - 1]);
      return r >= 2 && array_bytes / static_cast<unsigned long long>(r) + (1ull << 20) < 0xFFFFFFF0ull;
    };
    if (!lane_span_ok(ka) || !lane_span_ok(kb) || !store_tables_fit(ka, n)) return false;
  } else if (!strided_fits(ka, n2 * B, in_buf, a_in, BUF_SCRATCH, a_out) || !store_tables_fit(ka, n) ||
             !strided_fits(kb, B, BUF_SCRATCH, b_in, out_buf, b_out)) {
    return false;
  }
  scratch_bytes = std::max(scratch_bytes, need);
  stage sa = make_strided_stage(ka, outer * n2 * B, n2 * B, in_buf, a_in, BUF_SCRATCH, a_out, 1.0, backward, 1);
  sa.sa.stw_cdiv = B;
  sa.row_mode = 0;
  attach_store_tables(sa, n);
  out.push_back(sa);
  stage sb = make_strided_stage(kb, outer * n1 * B, B, BUF_SCRATCH, b_in, out_buf, b_out, scale, backward);
  if (outer > 1) {  // outer index of stage B = (array, k1): the array part advances by n * B on both sides
    sb.sa.outer_lo = n1;
    sb.sa.in_dist_outer_hi = n * B;
    sb.sa.out_dist_outer_hi = n * B;
  }
  out.push_back(sb);
  if (info != nullptr) {
    info->tier = PFFT_TIER_GLOBAL;
    info->n_factors = 2;
    info->factors[0] = static_cast<int>(n1);
    info->factors[1] = static_cast<int>(n2);
    info->workgroup_size = kb->wg;
    info->ffts_per_workgroup = kb->fpw;
    info->lds_bytes = std::max(ka->lds_bytes, kb->lds_bytes);
  }
  return true;
}

/// Plan `count` 1-D FFTs of length n.  Returns the tier used.
/// Three-stage plan of the GLOBAL tier for lengths whose two-factor split needs a factor above 2048 (N > 2^22: n = 4096
/// holds 4 fp32 columns -- 32-byte segments, 0.11 of peak at N = 2^23, 0.086 at 2^24): N = n1 * n2 * n3, the four-step
/// applied twice (reference: global_dispatcher.hpp:343-408 runs one kernel per factor of an arbitrary factor list).
///   S1  n1-point FFTs over stride n2 * n3 for every column c of [0, n2 * n3), x W_N^(k1 * c)        user in -> user out
///   S2  per row k1: n2-point FFTs over stride n3 for every column c3, x W_(n2 n3)^(k2 * c3)         user out -> scratch
///   S3  n3-point FFTs over c3 for every (k1, k2), result to X[k1 + n1 * k2 + n1 * n2 * k3]             scratch -> user out
/// S1 and S2 are ordinary stage-A launches.  S3's work-groups take t ADJACENT k1 (the index its output is contiguous
/// in), which are rows n2 * n3 apart after S2 -- so S2 writes the scratch as tiles [k1 % t][c3 % t] (blocks
/// [k1 / t][k2][c3 / t]: two-level outer index + group-major addressing of strided_args) and S3 reads each group's
/// t * n3 elements contiguously in its tiled-input form.  S2 / S3 run chunk by chunk like the two-stage plan.
bool plan_t::plan_three_stage(std::vector<stage>& out, long long n, long long count, const addressing& ia,
                              const addressing& oa, double scale, int backward, pfft_dim_info_t* info) {
  if (kn.no_three_stage || kn.no_precompiled ||
      kn.debug_global_set || kn.no_tiled_scratch ||
      kn.no_tiled_lanes) {
    return false;
  }
  long long min_n = (1ll << 22) + 1;  // beyond 2048 x 2048 a two-factor split needs n > 2048 (5 * 2^20: 0.121 against 0.222)
  if (kn.three_stage_min > 0) min_n = kn.three_stage_min;  // experiments
  if (n < min_n || static_cast<unsigned long long>(n) * elem_bytes() >= 0xFFFFFFF0ull) return false;
  const size_t per_transform = static_cast<size_t>(n) * elem_bytes();
  const bool cached = cache_chunk_bytes() >= per_transform &&
                      per_transform * static_cast<size_t>(count) >= cache_chunk_bytes() / 2;
  // SPLIT_COMPLEX user planes: the same three stages on runtime-specialised kernels -- S1 planes -> planes (in place on
  // the output planes), S2 planes -> interleaved scratch tiles, S3 tiles -> planes (mixed-storage forms, jit.cpp)
  const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
  if (split && !jit_enabled()) return false;
  // S3: a registered stage-B entry with whole-line groups and square tiles
  const strided_kernel* k3 = nullptr;
  long long n3 = 0;
  int t = 0;
  // fp32 up to 2^25: n3 = 256 first (all three stages on short kernels, several work-groups per CU: 2^23 0.232 -> 0.239,
  // 2^24 0.228 -> 0.237, 5 * 2^20 0.222 -> 0.233); fp64 and longer transforms: n3 = 1024 first (fp64 2^23 0.233 / 0.231, 2^26 equal)
  const long long want_n3 = kn.three_stage_n3;
  const bool short_first = desc.precision == PFFT_PRECISION_F32 && n <= (1ll << 25);
  const long long order[3] = {short_first ? 256ll : 1024ll, 512ll, short_first ? 1024ll : 256ll};
  for (long long len : order) {
    if (n % len != 0 || (want_n3 != 0 && len != want_n3)) continue;  // (PFFT_THREE_STAGE_N3: experiments)
    if (split) {
      std::string why;
      const strided_kernel* fb = jit_strided_kernel(desc.precision, len, 1024, false, 3, max_lds, &why, false, cached ? 2 : 0);
      if (fb == nullptr || fb->n_radices < 2 || (fb->fpw & (fb->fpw - 1)) != 0 || len % fb->fpw != 0 ||
          (len / fb->radices[0]) % fb->fpw != 0 || !jit_strided_ensure_mixed_tin(fb, &why)) {
        continue;
      }
      k3 = fb;
      n3 = len;
      t = fb->fpw;
      break;
    }
    const strided_kernel* fb = find_strided(len, false, false, -1, cached ? 2 : 0, false, 2, false);
    if (fb == nullptr) continue;
    const int tt = pair_tile(fb, len, false);
    if (tt == 0 || static_cast<size_t>(tt) * elem_bytes() < 128) continue;
    k3 = fb;
    n3 = len;
    t = tt;
    break;
  }
  if (k3 == nullptr) return false;
  // n1 * n2 = N / n3, n1 <= n2, t | n1 (tiles over k1); S2 needs a stage-A kernel of t columns
  const long long m12 = n / n3;
  long long n1 = 0;
  const strided_kernel* k2 = nullptr;
  bool k2_jit = false;
  for (long long c = static_cast<long long>(std::sqrt(static_cast<double>(m12))); c >= t && n1 == 0; --c) {
    if (m12 % c != 0 || c % t != 0) continue;
    const long long c2 = m12 / c;
    if (c2 > 2048 || strided_fpw(c, c2 * n3) <= 0) continue;
    const strided_kernel* fa = split ? nullptr : find_strided(c2, false, false, -1, cached ? 1 : 0, true, 1);
    if (fa != nullptr) {
      if (fa->fpw != t) continue;
      k2 = fa;
      k2_jit = false;
    } else {
      wg_params p;
      if (!jit_enabled() || !choose_strided_params(desc.precision, c2, n3, max_lds, &p, false, t) ||
          p.radices.size() < 2) {
        continue;
      }
      k2 = nullptr;
      k2_jit = true;
    }
    n1 = c;
  }
  if (n1 == 0) return false;
  const long long n2 = m12 / n1, M = n2 * n3;
  if (k2_jit) {
    std::string why;
    k2 = jit_strided_kernel(desc.precision, n2, n3, true, split ? 2 : 0, max_lds, &why, false, cached ? 1 : 0, t);
    if (k2 == nullptr || k2->fpw != t) return false;
  }
  const strided_kernel* k1 = nullptr;
  if (split) {
    std::string why;
    k1 = jit_strided_kernel(desc.precision, n1, M, true, 1, max_lds, &why);
    if (k1 == nullptr || k1->n_radices < 2) return false;
  } else {
    k1 = get_strided(n1, M, true, false, false, false, 0);
  }
  int sh = 0;
  while ((1 << sh) < t) ++sh;
  const addressing a1_in{ia.offset, M, 1, n}, a1_out{oa.offset, M, 1, n};
  const addressing a2_in{oa.offset, n3, 1, M}, a2_out{0, static_cast<long long>(t) * n3, 1, t};
  const addressing a3_in{0, 1, t, static_cast<long long>(t) * n3}, a3_out{oa.offset, n1 * n2, 1, n1};
  if (!strided_fits(k1, M, BUF_IN, a1_in, BUF_OUT, a1_out) || !store_tables_fit(k1, n) ||
      !strided_fits(k2, n3, BUF_OUT, a2_in, BUF_SCRATCH, a2_out) || !store_tables_fit(k2, M) ||
      !strided_fits(k3, n1, BUF_SCRATCH, a3_in, BUF_OUT, a3_out)) {
    return false;
  }
  long long chunk = static_cast<long long>((cached ? cache_chunk_bytes() : global_chunk_bytes()) / per_transform);
  chunk = even_chunks(std::max<long long>(1, std::min<long long>(chunk, count)), count);
  scratch_bytes = std::max(scratch_bytes, static_cast<size_t>(chunk) * per_transform);
  const int group_id = n_chunk_groups++;
  stage s1 = make_strided_stage(k1, count * M, M, BUF_IN, a1_in, BUF_OUT, a1_out, 1.0, backward, 1);
  attach_store_tables(s1, n);
  out.push_back(s1);
  stage s2 = make_strided_stage(k2, count * n1 * n3, n3, BUF_OUT, a2_in, BUF_SCRATCH, a2_out, 1.0, backward, 1, false);
  attach_store_tables(s2, M);
  {  // rows (b, k1) in, tiles [k1 % t][c3 % t] of the blocks [k1 / t][k2][c3 / t] out
    strided_args& a = s2.sa;
    a.outer_lo = t;
    a.in_dist_outer = M;
    a.in_dist_outer_hi = static_cast<long long>(t) * M;
    a.out_dist_outer = t;
    a.out_dist_outer_hi = static_cast<long long>(t) * M;
    a.out_gdist = static_cast<long long>(t) * t;
    a.out_stride = static_cast<unsigned>(static_cast<long long>(t) * n3);
    a.out_fdist = 1;
  }
  s2.chunk_group = group_id;
  s2.chunk_batches = chunk;
  s2.ffts_per_batch = n1 * n3;
  s2.in_batch_dist = n;
  s2.out_batch_dist = 0;
  if (k2->launch != nullptr && k2->fs_groups_per_wg > 0) s2.gpw = k2->fs_groups_per_wg;
  stage s3 = make_strided_stage(k3, count * n1 * n2, n1, BUF_SCRATCH, a3_in, BUF_OUT, a3_out, scale, backward, 0, false);
  {  // groups of t adjacent k1 for every (b, k2): t * n3 contiguous elements in, X[k1 + n1 * k2 + n1 * n2 * k3] out
    strided_args& a = s3.sa;
    a.outer_lo = n2;
    a.in_tile_shift = sh;
    a.in_stride = static_cast<unsigned>(t * t);
    a.in_fdist = static_cast<unsigned>(t);
    a.in_gdist = static_cast<long long>(t) * M;
    a.in_dist_outer = static_cast<long long>(t) * n3;
    a.in_dist_outer_hi = n;
    a.out_dist_outer = n1;
    a.out_dist_outer_hi = n;
  }
  s3.tiled_in = 1;
  s3.chunk_group = group_id;
  s3.chunk_batches = chunk;
  s3.ffts_per_batch = n1 * n2;
  s3.in_batch_dist = 0;
  s3.out_batch_dist = n;
  if (k3->fs_groups_per_wg > 0) s3.gpw = k3->fs_groups_per_wg;
  regrid_for_chunk(s2, std::min(chunk, count) * n1 * n3);
  regrid_for_chunk(s3, std::min(chunk, count) * n1 * n2);
  out.push_back(s2);
  out.push_back(s3);
  if (info != nullptr) {
    info->tier = PFFT_TIER_GLOBAL;
    info->n_factors = 3;
    info->factors[0] = static_cast<int>(n1);
    info->factors[1] = static_cast<int>(n2);
    info->factors[2] = static_cast<int>(n3);
    info->workgroup_size = k3->wg;
    info->ffts_per_workgroup = k3->fpw;
    info->lds_bytes = std::max(k1->lds_bytes, std::max(k2->lds_bytes, k3->lds_bytes));
  }
  return true;
}

/// XCC ids of the plan's device (census kernel, once per device and process); 0 when the census failed
int plan_t::xcd_queue_count() {
  static std::mutex m;
  static std::map<int, int> cache;
  std::lock_guard<std::mutex> lock(m);
  auto it = cache.find(device);
  if (it != cache.end()) return it->second;
  const int n = xcd_census(stream);
  cache[device] = n;
  return n;
}

/// GLOBAL tier, XCD-local form (stockham_xcd.hpp; the reference keeps its batches-in-flight inside the last-level cache,
/// committed_descriptor_impl.hpp:603-611, and runs one kernel per factor, dispatcher/global_dispatcher.hpp:343-408):
/// N = n1 x n2 with a registered pair runs as ONE persistent launch over the whole batch -- per-XCD task queues, stage
/// A of a transform and stage B of an earlier one side by side, the intermediate in per-XCD slot rings.  Taken only for
/// the pairs registered in kernels_xcd.hip (where the launch beat the two-launch plan on hardware) and batches that
/// fill its pipeline; PFFT_NO_XCD_LOCAL=1 keeps the two-launch plan (A/B twin of the parity tests).
bool plan_t::plan_xcd_local(std::vector<stage>& out, long long n, long long count, const addressing& ia,
                            const addressing& oa, double scale, int backward, pfft_dim_info_t* info) {
  if (desc.complex_storage != PFFT_INTERLEAVED_COMPLEX || kn.no_xcd_local ||
      kn.no_precompiled || kn.global_n1_set ||
      kn.debug_global_set) {
    return false;
  }
  int nk = 0;
  const xcd_kernel* ks = xcd_kernels(&nk);
  const xcd_kernel* k = nullptr;
  for (int i = 0; i < nk; ++i) {
    if (ks[i].precision == desc.precision && static_cast<long long>(ks[i].n1) * ks[i].n2 == n) k = &ks[i];
  }
  if (k == nullptr || static_cast<unsigned long long>(n) * elem_bytes() >= 0xFFFFFFF0ull || count >= (1ll << 27)) {
    return false;
  }
  const int n_queues = xcd_queue_count();
  if (n_queues <= 0) return false;
  // A queue needs transforms to run ahead of: below that the two launches win (measured: profiles/r4_xcd_local.md)
  // (the persistent launch has a fixed start-up; the crossovers are measured per entry, kernels_xcd.hip)
  long long min_batch = static_cast<long long>(std::max(24, 2 * k->slots)) * n_queues;
  min_batch = std::max<long long>(min_batch, (static_cast<long long>(k->min_mib) << 20) / (n * static_cast<long long>(elem_bytes())));
  if (kn.xcd_min_batch >= 0) min_batch = kn.xcd_min_batch;
  if (count < min_batch) return false;
  const long long n1 = k->n1, n2 = k->n2;
  const int t = k->fpw;
  int tsh = 0;
  while ((1 << tsh) < t) ++tsh;
  int slots = k->slots, lag = k->lag, lookahead = k->lookahead;
  if (kn.xcd_slots > 0) slots = kn.xcd_slots;  // schedule experiments
  if (kn.xcd_lag > 0) lag = kn.xcd_lag;
  if (slots < 2 || lag < 1 || lag >= slots || slots > 64) return false;
  // store-modifier tables W_N^(k1 * c) behind the kernel's own LDS (same shape rule as the two-launch stage A)
  strided_kernel shape{};
  shape.lds_bytes = k->stw_off + XCD_LDS_CTL_BYTES;
  shape.stw_mode = 1;
  int levels = 0, shift = 0;
  store_table_shape(&shape, n, &levels, &shift);
  const size_t stw_bytes = (static_cast<size_t>(levels) << shift) * elem_bytes();
  const size_t own = ((k->stw_off + stw_bytes + 15) & ~static_cast<size_t>(15)) + XCD_LDS_CTL_BYTES;
  if (levels > 4 || own > max_lds) return false;
  // the LDS request of a launch tuned for fewer work-groups per CU than would fit is padded until no more fit
  size_t lds = own;
  if (k->wg_per_cu > 0) {
    const size_t fits_one_more = (max_lds / static_cast<size_t>(k->wg_per_cu + 1) + 16 + 15) & ~static_cast<size_t>(15);
    if (fits_one_more <= max_lds / static_cast<size_t>(k->wg_per_cu)) lds = std::max(own, fits_one_more);
  }
  stage s;
  s.xcd = k;
  s.n = static_cast<int>(std::min<long long>(n, 0x7fffffff));
  s.in_buf = BUF_IN;
  s.out_buf = BUF_OUT;
  s.count = count;
  s.in_addr = ia;
  s.out_addr = oa;
  s.backward = backward;
  s.lds_bytes = lds;
  const void* tw_a = upload_twiddles(std::vector<int>(k->radices_a, k->radices_a + k->n_radices_a));
  const void* tw_b = upload_twiddles(std::vector<int>(k->radices_b, k->radices_b + k->n_radices_b));
  xcd_args& x = s.xa;
  // stage A: for every transform and column c: length-n1 FFT over rows (stride n2), x W_N^(k1 * c), group-major tiles out
  strided_args& a = x.a;
  a.tw = tw_a;
  a.twl_lds_off = k->twl_a_off;
  a.stw_lds_off = k->stw_off;
  a.total = count * n2;
  a.inner = n2;
  a.in_dist_outer = n;
  a.out_dist_outer = 0;  // (the kernel adds the slot's base)
  a.in_stride = static_cast<unsigned>(n2);
  a.in_fdist = 1;
  a.scale = 1.0;
  a.stw_tab = store_tables_for(n, levels, shift);
  a.stw_levels = levels;
  a.stw_lshift = shift;
  a.stw_cdiv = 1;
  a.out_gdist = n1 * t;
  a.out_stride = static_cast<unsigned>(t);
  a.out_fdist = 1;
  // stage B: for every transform and row k1: length-n2 FFT read from the tiles, output X[k1 + n1 * k2]
  strided_args& b = x.b;
  b.tw = tw_b;
  b.twl_lds_off = k->twl_b_off;
  b.stw_lds_off = k->stw_off;
  b.total = count * n1;
  b.inner = n1;
  b.in_dist_outer = 0;
  b.out_dist_outer = n;
  b.out_stride = static_cast<unsigned>(n1);
  b.out_fdist = 1;
  b.scale = scale;
  b.stw_cdiv = 1;
  b.in_tile_shift = tsh;
  b.in_stride = static_cast<unsigned>(n1 * t);
  b.in_fdist = static_cast<unsigned>(t);
  x.batch = count;
  x.n_queues = n_queues;
  x.slots = slots;
  x.lag = lag;
  x.lookahead = lookahead;
  // claim map: must outlast every ticket in flight -- slots + lag + lookahead batches plus three tickets per work-group
  const long long tpt = k->tasks_a + k->tasks_b;
  int per_cu = 0;
  hip_check(hipFuncSetAttribute(k->fn[backward], hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)),
            "hipFuncSetAttribute");
  // (the recovery launch that follows every execute asks for the same LDS: ADVICE r5)
  if (k->fn_recover[backward] != nullptr) {
    hip_check(hipFuncSetAttribute(k->fn_recover[backward], hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)),
              "hipFuncSetAttribute");
  }
  hip_check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k->fn[backward], k->wg, lds), "occupancy query");
  per_cu = std::max(per_cu, 1);
  s.grid = static_cast<unsigned>(per_cu * n_cus);
  // (sized for the largest grid the device could hold, not for this direction's: forward and backward share the block)
  const long long in_flight = 3ll * (8ll * n_cus) / std::max<long long>(tpt, 1) + slots + lag + 2 * lookahead + 8;
  int map_log2 = 6;
  while ((1ll << map_log2) < 2 * in_flight) ++map_log2;
  if (map_log2 > 14) return false;
  x.map_log2 = map_log2;
  x.max_iters = static_cast<unsigned>(std::min<long long>((count + lag + lookahead + 6) * tpt, 0xFFFFFFF0ll));
  if (kn.xcd_max_iters >= 0) x.max_iters = static_cast<unsigned>(kn.xcd_max_iters);  // tests: provoke a launch that gives up
  x.lds_ctl_off = static_cast<unsigned>(own - XCD_LDS_CTL_BYTES);
  x.prof = nullptr;
  const size_t ring = static_cast<size_t>(n_queues) * static_cast<size_t>(slots) * static_cast<size_t>(n) * elem_bytes();
  if (ring > global_chunk_bytes()) return false;
  scratch_bytes = std::max(scratch_bytes, ring);
  xcd_ctl_bytes = std::max(xcd_ctl_bytes, static_cast<size_t>(xcd_ctl_words(n_queues, slots, map_log2)) * sizeof(unsigned));
  xcd_tmap_bytes = std::max(xcd_tmap_bytes, static_cast<size_t>(count) * 8);
  // the recovery launch behind it: one work-group per ring slot at most (its private intermediate when it recomputes)
  s.recover_grid = static_cast<unsigned>(std::min<long long>(static_cast<long long>(n_queues) * slots, 2ll * n_cus));
  this->info.xcd_local[backward] = 1;
  out.push_back(s);
  if (info != nullptr) {
    info->tier = PFFT_TIER_GLOBAL;
    info->n_factors = 2;
    info->factors[0] = static_cast<int>(n1);
    info->factors[1] = static_cast<int>(n2);
    info->workgroup_size = k->wg;
    info->ffts_per_workgroup = k->fpw;
    info->lds_bytes = lds;
  }
  return true;
}

int plan_t::plan_1d(std::vector<stage>& out, long long n, long long count, long long inner_count, int in_buf,
                    const addressing& ia, int out_buf, const addressing& oa, bool packed_io, double scale,
                    int backward, pfft_dim_info_t* info) {
  const bool interleaved = desc.complex_storage == PFFT_INTERLEAVED_COMPLEX;
  if (info != nullptr) {
    info->length = static_cast<uint64_t>(n);
    info->n_factors = 0;
  }
  auto record = [&](int tier, const std::vector<int>& factors, int wg, int fpw, size_t lds) {
    if (info == nullptr) return;
    info->tier = tier;
    info->n_factors = static_cast<int>(std::min<size_t>(factors.size(), PFFT_MAX_FACTORS));
    for (int i = 0; i < info->n_factors; ++i) info->factors[i] = factors[static_cast<size_t>(i)];
    info->workgroup_size = wg;
    info->ffts_per_workgroup = fpw;
    info->lds_bytes = lds;
  };
  if (packed_io && (interleaved || (in_buf != BUF_SCRATCH && out_buf != BUF_SCRATCH))) {
    if (const spec_kernel* k = get_spec(n)) {
      out.push_back(make_spec_stage(k, count, in_buf, ia.offset, out_buf, oa.offset, scale, backward));
      record(k->n_radices == 1 ? PFFT_TIER_REGISTER : PFFT_TIER_WORKGROUP,
             std::vector<int>(k->radices, k->radices + k->n_radices), k->wg, k->fpw, k->lds_bytes);
      return PFFT_TIER_WORKGROUP;
    }
  }
  // UNPACKED layouts whose transforms do not interleave (padded rows, every k-th sample): the packed kernel's
  // configuration with runtime strides, lanes element-fastest
  {
    auto row_like = [&](const addressing& a) {
      return a.stride >= 1 && a.dist_inner >= (n - 1) * a.stride + 1 && a.stride < (1ll << 20) &&
             a.dist_inner < (1ll << 31);
    };
    const bool user_bufs = in_buf != BUF_SCRATCH && out_buf != BUF_SCRATCH;
    if (!packed_io && inner_count == count && row_like(ia) && row_like(oa) && (interleaved || user_bufs) &&
        !(ia.stride == 1 && ia.dist_inner == n && oa.stride == 1 && oa.dist_inner == n)) {
      const spec_kernel* k = find_spec(n, false);  // (no UNPACKED form of the register-resident entries)
      if (k == nullptr) {
        std::string why;
        k = jit_spec_kernel(desc.precision, n, !interleaved, max_lds, &why, true, nullptr, false);
      }
      auto fits = [&](const addressing& a) {
        const unsigned long long elems = static_cast<unsigned long long>(k->fpw - 1) * a.dist_inner +
                                         static_cast<unsigned long long>(n - 1) * a.stride + 1;
        return elems * elem_bytes() < 0xFFFFFFF0ull;
      };
      std::string why;
      const unpacked_kernel* u = (k != nullptr && fits(ia) && fits(oa)) ? jit_unpacked_kernel(k, !interleaved, &why)
                                                                        : nullptr;
      if (u != nullptr) {
        stage s = make_spec_stage(k, count, in_buf, ia.offset, out_buf, oa.offset, scale, backward, nullptr, u);
        s.in_addr = ia;
        s.out_addr = oa;
        out.push_back(s);
        record(k->n_radices == 1 ? PFFT_TIER_REGISTER : PFFT_TIER_WORKGROUP,
               std::vector<int>(k->radices, k->radices + k->n_radices), k->wg, k->fpw, k->lds_bytes);
        return PFFT_TIER_WORKGROUP;
      }
    }
  }
  // A batch-interleaved length whose full-width group (16 fp32 / 8 fp64 columns: whole 128-byte lines) does not fit the LDS
  // but fits the REGISTERS of one work-group -- fp32 1025 ... 2048 points, fp64 alike: up to 256 KiB, half of it as the LDS image --
  // runs in ONE HBM pass on the register-resident strided kernel, one work-group per CU (stockham_strided_hx.hpp; jit_strided_kernel
  // with a negative group width: that kernel or nothing -- a plan whose kernel needs scratch hands the length back).  Before
  // round 6's last day these lengths took the two-stage plan below (two passes) or, in split storage, 8-column groups.
  auto plan_wide_group_of = [&](long long inner, int full_fpw) {
    if (!jit_enabled() || kn.no_bi_wide || inner < full_fpw || strided_fpw(n, inner) >= full_fpw) return false;
    // (split storage: a plane's row pitch is inner * sizeof(scalar))
    const bool unal = !kn.no_unaligned_policy && (interleaved || !kn.no_split_unaligned_policy) &&
                      (static_cast<unsigned long long>(inner) * (interleaved ? elem_bytes() : elem_bytes() / 2)) % 128 != 0 &&
                      static_cast<unsigned long long>(n) * static_cast<unsigned long long>(count) * elem_bytes() >= (64ull << 20);
    // an array of 4 GiB and more (a 1-D batch-interleaved descriptor, interleaved storage): the BIG form of the same kernel (64-bit
    // butterfly-leg offsets, stockham_strided.hpp) under the conditions of the two-stage plan's BIG forms -- what stays 32 bits is a
    // lane's offset, the span of the first / last pass's butterflies = 1 / radix of the array
    const unsigned long long array_bytes = static_cast<unsigned long long>(n) * static_cast<unsigned long long>(inner) * elem_bytes();
    const bool big = array_bytes >= 0xFFFFFFF0ull;
    if (big && (desc.rank != 1 || !interleaved || kn.no_big_bi || array_bytes > (16ull << 30) ||
                static_cast<unsigned long long>(n) * static_cast<unsigned long long>(inner) >= (1ull << 32))) {
      return false;
    }
    std::string why;
    const strided_kernel* k = jit_strided_kernel(desc.precision, n, inner, false, interleaved ? 0 : 1, max_lds, &why, true,
                                                 unal ? 3 : 0, -full_fpw, big);
    if (k == nullptr) {
      jit_note("strided (wide group)", n, why);
      return false;
    }
    if (big) {
      const int r = k->n_radices >= 1 ? std::min(k->radices[0], k->radices[k->n_radices - 1]) : 0;
      if (r < 2 || array_bytes / static_cast<unsigned long long>(r) + (1ull << 20) >= 0xFFFFFFF0ull) return false;
    } else if (!strided_fits(k, inner_count, in_buf, ia, out_buf, oa)) {
      return false;
    }
    out.push_back(make_strided_stage(k, count, inner_count, in_buf, ia, out_buf, oa, scale, backward));
    record(PFFT_TIER_WORKGROUP, std::vector<int>(k->radices, k->radices + k->n_radices), k->wg, k->fpw, k->lds_bytes);
    return true;
  };
  // SPLIT_COMPLEX planes hold 4 / 8 bytes per element: a whole 128-byte line per plane takes 32 fp32 / 16 fp64 columns, a group the
  // LDS holds up to N = 512 only.  N = 513 ... 1024 in split storage therefore go to the wide kernel at DOUBLE width first (<= 32 values per
  // lane on 1024 / 512 lanes; profiles/r6_bi_wide_split32.txt, 16 / 8 columns LDS-resident -> 32 / 16 register-resident, fraction of the HBM
  // peak): fp32 576 0.357 -> 0.490, 640 0.350 -> 0.528, 768 0.357 -> 0.568, 896 0.355 -> 0.519, 1024 0.369 -> 0.446; fp64 0.336 -> 0.540, 0.332 -> 0.535,
  // 0.379 -> 0.588, 0.375 -> 0.541, 0.321 -> 0.537.  (Interleaved data at that width: 640 +13 ... 18 %, 768 +2 ... 5 %, 1024 -3 ... -8 % -- not taken.)
  auto plan_wide_group = [&](long long inner) {
    const int full_fpw = desc.precision == PFFT_PRECISION_F64 ? 8 : 16;
    if (kn.bi_wide_fpw > 0) return plan_wide_group_of(inner, kn.bi_wide_fpw);  // experiments (PFFT_BI_WIDE_FPW): that group width
    if (!interleaved && !kn.no_bi_wide_split2 && plan_wide_group_of(inner, 2 * full_fpw)) return true;
    return plan_wide_group_of(inner, full_fpw);
  };
  // Long batch-interleaved transforms: one work-group could hold only a few columns (narrow HBM segments), so
  // split N = n1 * n2 and run both four-step stages column shaped with full-width groups, through scratch.
  if (desc.rank == 1 && in_buf == BUF_IN && out_buf == BUF_OUT && ia.dist_inner == 1 &&
      oa.dist_inner == 1 && ia.stride == count && oa.stride == count && inner_count == count) {
    if (plan_wide_group(count)) return PFFT_TIER_WORKGROUP;
    if (interleaved && plan_batch_interleaved_two_stage(out, n, count, 1, in_buf, out_buf, ia, oa, scale, backward, info)) {
      return PFFT_TIER_GLOBAL;
    }
  }
  // ... and long column dimensions of N-D arrays: `inner_count` adjacent columns per array, arrays n * inner apart
  if (desc.rank > 1 && in_buf != BUF_SCRATCH && out_buf != BUF_SCRATCH && ia.dist_inner == 1 &&
      oa.dist_inner == 1 && ia.stride == inner_count && oa.stride == inner_count && inner_count > 0 &&
      count % inner_count == 0 && ia.dist_outer == n * inner_count && oa.dist_outer == n * inner_count &&
      !kn.nd_two_stage_columns) {
    if (plan_wide_group(inner_count)) return PFFT_TIER_WORKGROUP;
    if (interleaved && plan_batch_interleaved_two_stage(out, n, inner_count, count / inner_count, in_buf, out_buf, ia, oa, scale,
                                                        backward, info)) {
      return PFFT_TIER_GLOBAL;
    }
  }
  // the strided tier pays when at least one side is "column" shaped (consecutive FFTs adjacent in memory)
  const bool column_shaped = ia.dist_inner == 1 || oa.dist_inner == 1;
  const bool user_split = !interleaved && in_buf != BUF_SCRATCH;
  const bool column_both = ia.dist_inner == 1 && oa.dist_inner == 1;
  const bool row_side = (ia.stride == 1 && ia.dist_inner != 1) || (oa.stride == 1 && oa.dist_inner != 1);
  // a column-shaped OUTPUT whose row pitch is no multiple of a 128-byte line (a batch-interleaved layout with a batch count that
  // is no multiple of 16 fp32 / 8 fp64 transforms): policy 3, kernels.hpp aux_of_policy.  The partial-line STORES are what
  // costs -- P -> BI N = 1024 x 131 077 0.232 -> 0.404, N = 256 0.353 -> 0.533, BI -> BI 0.204 -> 0.497; an unaligned INPUT alone (BI -> P) is
  // -9 ... +16 % either way and keeps the streamed kernels (profiles/r6_pbi_policy.txt)
  // (split user planes: a plane's pitch is stride * sizeof(scalar); batch-interleaved on both sides only -- measured on the wide groups,
  //  profiles/r6_bi_wide_split_unaligned.txt, and on N = 256 / 512, r6_bi_split_unaligned_small.txt)
  auto unaligned = [&](const addressing& a) {
    const size_t eb = interleaved ? elem_bytes() : elem_bytes() / 2;
    return a.dist_inner == 1 && a.stride > 1 && (static_cast<unsigned long long>(a.stride) * eb) % 128 != 0;
  };
  // (from 64 MiB of data: below that a commit does not pay a compilation for a pre-compiled length, and the data sits in the
  //  caches whatever the policy)
  const bool worth = static_cast<unsigned long long>(count) * static_cast<unsigned long long>(n) * elem_bytes() >= (64ull << 20);
  const int stage_policy =
      (tail_policy == 0 && (interleaved || (user_split && column_both && !kn.no_split_unaligned_policy)) && worth &&
       !kn.no_unaligned_policy && unaligned(oa))
          ? 3
          : tail_policy;
  if (const strided_kernel* k =
          column_shaped ? get_strided(n, inner_count, false, user_split, column_both, row_side, stage_policy) : nullptr;
      strided_fits(k, inner_count, in_buf, ia, out_buf, oa)) {
    out.push_back(make_strided_stage(k, count, inner_count, in_buf, ia, out_buf, oa, scale, backward));
    record(PFFT_TIER_WORKGROUP, std::vector<int>(k->radices, k->radices + k->n_radices), k->wg, k->fpw,
           k->lds_bytes);
    return PFFT_TIER_WORKGROUP;
  }
  if (n <= generic_max_n()) {
    stage s = make_generic_stage(n, count, inner_count, in_buf, ia, out_buf, oa, scale, backward, backward);
    record(PFFT_TIER_GENERIC, std::vector<int>(s.ga.radix, s.ga.radix + s.ga.n_passes), GENERIC_WG, s.ga.fpw,
           s.lds_bytes);
    out.push_back(s);
    return PFFT_TIER_GENERIC;
  }
  // ---- GLOBAL tier: N = N1 * N2 through HBM scratch (four-step) ----
  // Like the reference (committed_descriptor_impl.hpp:757-764) only for 1-D packed data.
  if (!packed_io || desc.rank != 1) {
    fail(PFFT_UNSUPPORTED_CONFIGURATION, "FFT size ", n,
         " needs the multi-kernel (global) implementation, which is only supported for 1-D transforms in the "
         "default (packed) layout");
  }
  if (in_buf == BUF_IN && out_buf == BUF_OUT && plan_three_stage(out, n, count, ia, oa, scale, backward, info)) {
    return PFFT_TIER_GLOBAL;
  }
  if (in_buf == BUF_IN && out_buf == BUF_OUT && plan_xcd_local(out, n, count, ia, oa, scale, backward, info)) {
    return PFFT_TIER_GLOBAL;
  }
  const long long gmax = generic_max_n();
  long long n1 = 0;
  // most balanced split whose two lengths both have a strided work-group kernel ...
  for (long long c = static_cast<long long>(std::sqrt(static_cast<double>(n))); c >= 2; --c) {
    if (n % c == 0 && strided_fpw(c, n / c) > 0 && strided_fpw(n / c, c) > 0) {
      n1 = c;
      break;
    }
  }
  long long want_n1 = forced_n1;
  if (kn.global_n1_set) want_n1 = kn.global_n1;  // experiments: force the first factor of the split
  if (want_n1 >= 2 && n % want_n1 == 0 && strided_fpw(want_n1, n / want_n1) > 0 && strided_fpw(n / want_n1, want_n1) > 0) {
    n1 = want_n1;
  } else {
    want_n1 = 0;
  }
  // ... otherwise the most balanced split whose two lengths both run on the generic tier
  for (long long c = static_cast<long long>(std::sqrt(static_cast<double>(n))); n1 == 0 && c >= 2; --c) {
    if (n % c == 0 && n / c <= gmax && !choose_radices(c).empty() && !choose_radices(n / c).empty()) {
      n1 = c;
      break;
    }
  }
  if (n1 == 0) {
    fail(PFFT_UNSUPPORTED_CONFIGURATION, "FFT size ", n, " cannot be split into two factors that fit local memory",
         " (large prime factors are not supported)");
  }
  long long n2 = n / n1;
  bool paired_split = false;
  // Stage pairs (below) need a registered stage-B entry for n2 and, for n1, a registered stage-A entry or a
  // runtime-specialised kernel of the same group width.  Among the splits that allow one, a SHORT stage A wins over
  // a balanced split -- several stage-A work-groups per CU, stage B on the best-tuned entries (n2 = 1024 / 512):
  // measured (tools/probes/half_pairs_n1.sh, fraction of peak) fp32 3 * 2^17: 384 x 1024 0.313, 768 x 512 0.294,
  // 512 x 768 without a pair 0.261; 3 * 2^16: 192 x 1024 0.355, 384 x 512 0.310; 5 * 2^15: 160 x 1024 0.347, 320 x 512
  // 0.275; 2^18: 256 x 1024 0.364, 512 x 512 0.347; fp64 3 * 2^16: 0.372 against 0.326 -- but not a very short one
  // (3 * 2^15: 192 x 512 0.354, 96 x 1024 0.340; 2^16: 256 x 256 0.379, 128 x 512 0.367, 64 x 1024 0.316; 2^17: 256 x 512
  // 0.370, 128 x 1024 0.357; in fp64 n1 = 128 still wins: 2^16 0.386 against 0.371, 2^17 0.378 against 0.356).  So: the
  // smallest n1 >= 160 (fp64: 128) that pairs, else the largest below.  A registered length without a stage-A entry
  // of stage B's width (n1 = 128, 64) gets a runtime-specialised stage A like any other (2^15 as 128 x 256 with it
  // 0.375, on the registered 32-column entry 0.358; fp64 0.396 / 0.370).  A stage A narrower than a 128-byte line
  // never comes out of this search (5 * 2^18 as 640 x 2048 on 8 columns: 0.224 against 0.242), and stage-B entries whose
  // own output segments are that narrow (n2 = 2048 -- their tiles may still be a line wide, pair_tile) rank last.
  if (desc.complex_storage == PFFT_INTERLEAVED_COMPLEX && jit_enabled() && want_n1 == 0 &&
      !kn.no_fs_pairs && !kn.no_half_pairs &&
      !kn.no_tiled_scratch && !kn.no_tiled_lanes &&
      !kn.no_precompiled && !kn.debug_global_set) {
    // 0: no pair; 1: pairs, stage B's own groups span whole lines; 2: pairs, but stage B's OUTPUT segments are
    // narrower than a line (fp32 n2 = 2048: 8 columns -- its tiles may still be 16 wide, pair_tile)
    auto pairable = [&](long long m, long long len) {  // m: stage A's length, len: stage B's
      const strided_kernel* fb = find_strided(len, false, false, -1, 0, false, 2);
      if (fb == nullptr) return 0;
      for (int wide = 1; wide >= 0; --wide) {
        const int t = pair_tile(fb, len, wide != 0);
        if (t == 0 || static_cast<size_t>(t) * elem_bytes() < 128) continue;  // stage A in whole lines only
        const int kind = static_cast<size_t>(fb->fpw) * elem_bytes() < 128 ? 2 : 1;
        if (const strided_kernel* fa = find_strided(m, false, false, -1, 0, true, 1)) {
          if (fa->fpw == t) return kind;
          continue;
        }
        wg_params p;
        if (choose_strided_params(desc.precision, m, len, max_lds, &p, false, t) && p.radices.size() >= 2) return kind;
      }
      return 0;
    };
    int count_k = 0;
    const strided_kernel* k =
        desc.precision == PFFT_PRECISION_F64 ? strided_kernels_f64(&count_k) : strided_kernels_f32(&count_k);
    const long long short_a = desc.precision == PFFT_PRECISION_F64 ? 128 : 160;
    // [0]: candidates whose stage B writes whole lines, [1]: the others (taken only when [0] is empty);
    // per class: the smallest pairing n1 >= short_a, the largest pairing n1 below
    long long above_c[2] = {0, 0}, below_c[2] = {0, 0};
    for (int i = 0; i < count_k; ++i) {
      const long long len = k[i].n;
      if (k[i].fs_b == 0 || k[i].policy != 0 || n % len != 0 || n / len < 2) continue;
      const long long m = n / len;
      if (strided_fpw(m, len) <= 0) continue;
      const int kind = pairable(m, len);
      if (kind == 0) continue;
      long long& above = above_c[kind - 1];
      long long& below = below_c[kind - 1];
      if (m >= short_a && (above == 0 || m < above)) above = m;
      if (m < short_a && m > below) below = m;
    }
    const int cls = (above_c[0] != 0 || below_c[0] != 0) ? 0 : 1;
    const long long above = above_c[cls], below = below_c[cls];
    if (above != 0 || below != 0) {
      n1 = above != 0 ? above : below;
      n2 = n / n1;
      paired_split = true;
    }
  }
  // No registered entry pairs with any factor (10^5, 68640 = 2^5 3 5 11 13, ...): both stages are runtime-specialised and
  // the balanced split is the worst shape for them -- two mid-sized stages, each alone on its CU behind three barriers.
  // Measured over every divisor (tools/probes/split_sweep.py, fp32, fraction of the HBM peak, balanced -> best):
  // 30000 0.247 -> 0.287, 40000 0.263 -> 0.307, 62500 0.229 -> 0.266, 68640 0.221 -> 0.299, 10^5 0.176 -> 0.270,
  // 120000 0.158 -> 0.290, 250000 0.138 -> 0.256; a LONG stage A (400 ... 1000 points) in front of a SHORT stage B (60 ... 256)
  // is at or within 10 % of the best of every one of them, n1 = 500 in front of n2 = 60 ... 240 at the very top of five:
  // the n1 closest to 500 (from below rather than from above) with n2 in 60 ... 256.
  // fp64 has no such pattern (68640: 260 x 264 0.373, 156 x 440 0.382, 480 x 143 0.282; 10^5: 500 x 200 0.364, 250 x 400
  // 0.287): PFFT_PLAN_MEASURE=1 times the candidates instead (measured_split).
  if (!paired_split && desc.precision == PFFT_PRECISION_F32 && desc.complex_storage == PFFT_INTERLEAVED_COMPLEX &&
      jit_enabled() && want_n1 == 0 && !kn.no_split_rule &&
      !kn.debug_global_set) {
    long long best = 0;
    double best_d = 0;
    for (long long c = 384; c <= 1024; ++c) {
      if (n % c != 0) continue;
      const long long m = n / c;
      if (m < 60 || m > 256 || strided_fpw(c, m) <= 0 || strided_fpw(m, c) <= 0) continue;
      // (above 500 the distance counts three times: 68640 as 480 x 143 0.285, as 520 x 132 0.250)
      const double d = std::fabs(std::log(static_cast<double>(c) / 500.0)) * (c > 500 ? 3.0 : 1.0);
      if (best == 0 || d < best_d) {
        best = c;
        best_d = d;
      }
    }
    if (best != 0) {
      n1 = best;
      n2 = n / best;
    }
  }
  if (want_n1 == 0 && jit_enabled() && !kn.debug_global_set &&
      !kn.no_split_rule) {  // the tuned table of this architecture (pairs included: it is measured)
    const std::vector<int> tuned = builtin_choice(jit_device_arch(), desc.precision, n, true);
    if (tuned.size() == 2 && strided_fpw(tuned[0], tuned[1]) > 0 && strided_fpw(tuned[1], tuned[0]) > 0) {
      n1 = tuned[0];
      n2 = tuned[1];
    }
  }
  if (want_n1 == 0 && plan_measure_enabled() && jit_enabled() && in_buf == BUF_IN && out_buf == BUF_OUT &&
      !kn.debug_global_set) {
    n1 = measured_split(n, count, n1);
    n2 = n / n1;
  }
  // Chunking (the reference's num_batches_in_l2 idea, committed_descriptor_impl.hpp:603-611) bounds the scratch.
  // Measured on MI355X (profiles/r1_notes.md): cache-sized chunks (16-256 MiB) do NOT make stage B's reads hit the
  // Infinity Cache -- they only shrink the launches -- so the default chunk is as large as the scratch cap allows.
  // Round 2: with the intermediate of a chunk written by default-policy stores and read by default-policy loads
  // (everything else streamed) a chunk of the Infinity Cache's size IS served on-die (cache_chunk_bytes()).
  const size_t per_transform = static_cast<size_t>(n) * elem_bytes();
  const bool interleaved_io = desc.complex_storage == PFFT_INTERLEAVED_COMPLEX;
  // Measured with random data (tools/perf_cache.py): fp32 N=65536 x 2048 +5.5 %, fp32 2^20 x 256 +3.3 %, fp64 65536 x
  // 512 +7 %; below 128 MiB of intermediate the streamed kernels win (-8 % at 64 MiB), and a batch that needs
  // several chunks of one-work-group-per-CU kernels (C3: fp64 1024-point stages, 130 KiB of LDS) loses ~1 % to the
  // tails of the extra launches, so those two cases keep round 1's plan.
  const size_t all_bytes = per_transform * static_cast<size_t>(count);
  // (SPLIT_COMPLEX user data: the mixed-storage stage kernels carry the same writer / reader policies; PFFT_SPLIT_CACHED=0
  //  keeps them streamed and unchunked as in round 2)
  const bool split_cached = kn.split_cached;
  bool cached = (interleaved_io || split_cached) && cache_chunk_bytes() >= per_transform &&
                all_bytes >= cache_chunk_bytes() / 2;
  if (cached && all_bytes > cache_chunk_bytes()) {
    const strided_kernel* pa = find_strided(n1);
    const strided_kernel* pb = find_strided(n2);
    const size_t big = 80 * 1024;
    // (with the chunks overlapped -- chunk_overlap_mode() 2 -- those plans gain too: C3 1.652 -> 1.592 ms)
    if (((pa != nullptr && pa->lds_bytes > big) || (pb != nullptr && pb->lds_bytes > big)) &&
        overlap_mode != 2) {
      cached = false;
    }
  }
  long long chunk = static_cast<long long>((cached ? cache_chunk_bytes() : global_chunk_bytes()) / per_transform);
  chunk = even_chunks(std::max<long long>(1, std::min<long long>(chunk, count)), count);
  const int group_id = n_chunk_groups++;
  // stage A: for every batch b and column c: length-n1 FFT over rows (stride n2), x W_n^{k1*c}, same layout out
  addressing a_in{ia.offset, n2, 1, n};
  addressing a_out{0, n2, 1, n};
  const bool interleaved_user = desc.complex_storage == PFFT_INTERLEAVED_COMPLEX;
  const bool user_io = in_buf != BUF_SCRATCH && out_buf != BUF_SCRATCH;
  // (the default kernels of the two lengths are fetched -- and, for unregistered lengths, compiled -- only when no
  //  pair takes their place: default_kernels below)
  const strided_kernel* ka = nullptr;
  const strided_kernel* kb = nullptr;
  // Four-step pair: entries tuned as stage A / stage B of a group-major intermediate with equal group widths
  // (strided_kernel::fs_a / fs_b; PFFT_NO_FS_PAIRS=1 keeps the default entries of the two lengths)
  bool fs_pair = false;
  if (interleaved_user && !kn.no_fs_pairs && !kn.no_tiled_scratch &&
      !kn.no_tiled_lanes && !kn.no_precompiled) {
    const strided_kernel* fa = find_strided(n1, false, false, -1, cached ? 1 : 0, true, 1);
    const strided_kernel* fb = find_strided(n2, false, false, -1, cached ? 2 : 0, false, 2);
    if (fa != nullptr && fb != nullptr &&
        (fa->fpw == pair_tile(fb, n2, false) || fa->fpw == pair_tile(fb, n2, true)) &&
        static_cast<unsigned long long>(n) * elem_bytes() < 0xFFFFFFF0ull && store_tables_fit(fa, n)) {
      ka = fa;
      kb = fb;
      fs_pair = true;
    }
  }
  // Half pair: only stage B's length has a registered entry (N = 3 * 2^18 = 768 x 1024, 5 * 2^17 = 640 x 1024, ...):
  // stage A is runtime-specialised with stage B's group width, writes the group-major intermediate, and stage B
  // reads it in its tiled-input form instead of row-staging a row-major one (PFFT_NO_HALF_PAIRS=1: round-3 plan)
  bool half_pair = false;
  if (!fs_pair && interleaved_user && jit_enabled() && !kn.no_fs_pairs &&
      !kn.no_half_pairs && !kn.no_tiled_scratch &&
      !kn.no_tiled_lanes && !kn.no_precompiled &&
      !kn.debug_global_set && find_strided(n1, false, false, -1, 0, true, 1) == nullptr) {
    const strided_kernel* da = find_strided(n1);  // (a registered default of the pairing width pairs by itself below)
    for (int pass = 0; pass < 4 && !half_pair; ++pass) {
      const bool with_ltw = pass < 2, wide = (pass & 1) == 0;
      const strided_kernel* fb = find_strided(n2, false, false, -1, cached ? 2 : 0, false, 2, with_ltw);
      if (fb == nullptr || static_cast<unsigned long long>(n) * elem_bytes() >= 0xFFFFFFF0ull) continue;
      const int t = pair_tile(fb, n2, wide);
      if (t == 0 || (da != nullptr && da->fpw == t)) continue;
      const bool on_loads = fb->fs_ltw != 0;
      if (on_loads && (!with_ltw || wide || fb->stw_mode != 1 || !store_tables_fit(fb, n))) continue;
      wg_params probe;
      if (!choose_strided_params(desc.precision, n1, n2, max_lds, &probe, false, t) || probe.radices.size() < 2) continue;
      std::string why;
      const strided_kernel* fa =
          jit_strided_kernel(desc.precision, n1, n2, !on_loads, 0, max_lds, &why, false, cached ? 1 : 0, t);
      if (fa == nullptr || fa->fpw != t || fa->n_radices < 2 || (!on_loads && !store_tables_fit(fa, n)) ||
          !strided_fits(fa, n2, in_buf, addressing{ia.offset, n2, 1, n}, BUF_SCRATCH, addressing{0, n2, 1, n})) {
        continue;
      }
      ka = fa;
      kb = fb;
      fs_pair = half_pair = true;
    }
  }
  if (!fs_pair) {  // default_kernels
    ka = interleaved_user ? get_strided(n1, n2, true, false, false, false, cached ? 1 : 0)
                          : (user_io ? get_strided_mixed(n1, n2, 2, cached ? 1 : 0) : nullptr);
    kb = interleaved_user ? get_strided(n2, n1, false, false, false, true, cached ? 2 : 0)  // rows in
                          : (user_io ? get_strided_mixed(n2, n1, 3, cached ? 2 : 0) : nullptr);
  }
  // the pair's stage B may carry the inter-stage twiddles on its loads; stage A then has no store modifier
  // (PFFT_NO_LTW=1: the modifier stays on stage A's stores)
  bool ltw = fs_pair && kb->fs_ltw != 0 && kb->stw_mode == 1 && store_tables_fit(kb, n) &&
             (half_pair || ka->fn[0] != nullptr) && !kn.no_ltw &&
             !kn.debug_global_set &&
             strided_fits(ka, n2, in_buf, addressing{ia.offset, n2, 1, n}, BUF_SCRATCH, addressing{0, n2, 1, n});
  // (Round 6 measured both extensions of this and adopted neither, profiles/r6_stage_a_hx_and_jit_overlap.txt: the overlap for
  //  plans whose stage A is compiled at commit LOSES -- 68640 0.307 -> 0.277, 62500 0.278 -> 0.248, 10^6 0.256 -> 0.245 --, and the
  //  register-resident form of the registered one-per-CU stage A of a pair changes nothing: C3 0.362-0.364 against 0.358-0.360.)
  // Scratch: one chunk -- or two halves that alternate when consecutive chunks overlap (the first launch of chunk
  // c + 1 without the in-order barrier).  Only a pre-compiled interleaved stage A can launch that way (pfa_launch);
  // plans on mixed-storage, runtime-compiled or generic stages keep ONE buffer, and a plan whose chunk is the scratch
  // cap itself halves the chunk instead of doubling the allocation.
  {
    const bool any_order_capable = chunk < count && chunk_overlap_enabled() && interleaved_user && ka != nullptr &&
                                   ka->launch != nullptr && kb != nullptr;
    if (any_order_capable && !cached && 2 * static_cast<size_t>(chunk) * per_transform > global_chunk_bytes()) {
      chunk = std::max<long long>(1, chunk / 2);
    }
    const size_t need = static_cast<size_t>(chunk) * per_transform;
    scratch_bytes = std::max(scratch_bytes, need);
    if (any_order_capable) {
      overlap_scratch_half = std::max(overlap_scratch_half, need);
      scratch_bytes = std::max(scratch_bytes, 2 * overlap_scratch_half);
    }
  }
  // PFFT_DEBUG_GLOBAL, debugging aid: "ga" / "gb" force the generic kernel for a stage
  const bool force_generic_a = kn.debug_global.find("ga") != std::string::npos;
  const bool force_generic_b = kn.debug_global.find("gb") != std::string::npos;
  stage sa;
  if (!force_generic_a && strided_fits(ka, n2, in_buf, a_in, BUF_SCRATCH, a_out) && (ltw || store_tables_fit(ka, n))) {
    // conjugating on load and store in both stages is the identity in between, so the backward transform can use
    // the kernels' BWD form on both
    sa = make_strided_stage(ka, count * n2, n2, in_buf, a_in, BUF_SCRATCH, a_out, 1.0, backward, ltw ? 0 : 1);
    if (!ltw) attach_store_tables(sa, n);
    if (fs_pair && ka->fs_groups_per_wg > 0) sa.gpw = ka->fs_groups_per_wg;
    // a runtime-specialised stage A that is alone on its CU takes eight groups per work-group like the registered
    // n = 1024 entries (its twiddle / modifier tables are copied to LDS once per work-group): 3 * 2^18 fp32 0.293 ->
    // 0.303, fp64 0.314 -> 0.332; the short ones (several per CU) are indifferent or lose 1-2 %
    if (half_pair && ka->launch == nullptr && ka->lds_bytes > 80 * 1024) sa.gpw = 8;
  } else {
    ltw = false;  // (cannot happen for a registered pair; the generic stage A always carries the modifier itself)
    sa = make_generic_stage(n1, count * n2, n2, in_buf, a_in, BUF_SCRATCH, a_out, 1.0, backward, backward);
    int shift = 0;  // the generic kernel reads two global tables (hi/lo split of the exponent)
    while ((1ll << (2 * shift)) < n) ++shift;
    const void* stw_lo = nullptr;
    const void* stw_hi = nullptr;
    upload_store_twiddles(n, shift, &stw_lo, &stw_hi);
    sa.ga.stw_lo = stw_lo;
    sa.ga.stw_hi = stw_hi;
    sa.ga.stw_shift = shift;
  }
  sa.chunk_group = group_id;
  sa.chunk_batches = chunk;
  sa.ffts_per_batch = n2;
  sa.in_batch_dist = n;
  sa.out_batch_dist = 0;
  out.push_back(sa);
  // stage B: for every batch b and row k1: length-n2 FFT (contiguous), output X[k1 + n1*k2]
  addressing b_in{0, 1, n2, n};
  addressing b_out{oa.offset, n1, 1, n};
  stage sb;
  // SPLIT_COMPLEX user data: when both mixed-storage stage kernels hold the same number of columns, the intermediate
  // is group-major too and stage B reads it in its tiled-input form (scratch tiles -> the user's planes) instead of
  // row-staging a row-major one (PFFT_NO_SPLIT_TILED=1: round-3 plan)
  bool split_tiled = false;
  if (!interleaved_user && user_io && ka != nullptr && kb != nullptr && ka->launch == nullptr &&
      kb->launch == nullptr && ka->fpw == kb->fpw && (kb->fpw & (kb->fpw - 1)) == 0 && kb->n_radices >= 2 &&
      ka->n_radices >= 2 && n2 % kb->fpw == 0 && (n2 / kb->radices[0]) % kb->fpw == 0 &&
      static_cast<unsigned long long>(n) * elem_bytes() < 0xFFFFFFF0ull && !force_generic_a && !force_generic_b &&
      sa.strided == ka && !kn.no_split_tiled && !kn.no_tiled_scratch &&
      !kn.no_tiled_lanes) {
    std::string why;
    split_tiled = jit_strided_ensure_mixed_tin(kb, &why);
  }
  if (!force_generic_b && strided_fits(kb, n1, BUF_SCRATCH, b_in, out_buf, b_out)) {
    sb = make_strided_stage(kb, count * n1, n1, BUF_SCRATCH, b_in, out_buf, b_out, scale, backward, 0,
                            !fs_pair && !split_tiled);
    if (fs_pair && kb->fs_groups_per_wg > 0) sb.gpw = kb->fs_groups_per_wg;
  } else {
    sb = make_generic_stage(n2, count * n1, n1, BUF_SCRATCH, b_in, out_buf, b_out, scale, backward, backward);
  }
  sb.chunk_group = group_id;
  sb.chunk_batches = chunk;
  sb.ffts_per_batch = n1;
  sb.in_batch_dist = 0;
  sb.out_batch_dist = n;
  // Group-major intermediate: stage A's work-group (FPW_A adjacent columns, all n1 rows) writes its n1 x FPW_A
  // block contiguously; stage B then reads row k1 as n2 / FPW_A tiles of FPW_A elements, and its FPW_B adjacent rows
  // share FPW_A * FPW_B contiguous elements per tile.  Measured on the C3 stages (tools/tune_strided.hip): column
  // kernel with a contiguous instead of a strided side 4.9 -> 5.3 (output) / 5.8 (input) TB/s.
  if (sa.strided != nullptr && sb.strided != nullptr && sb.row_mode == 0 && !kn.no_tiled_scratch) {
    const int t = sa.strided->fpw;
    int sh = 0;
    while ((1 << sh) < t) ++sh;
    const long long nb0 = n2 / sb.strided->radices[0];
    if ((1 << sh) == t && n2 % t == 0 && nb0 % t == 0 &&
        static_cast<unsigned long long>(n) * elem_bytes() < 0xFFFFFFF0ull) {
      // PFFT_GLOBAL_LAYOUT=b (experiment): intermediate contiguous per stage-B work-group instead of per stage-A
      // work-group.  Measured on C3: 1.675-1.702 ms against 1.660-1.668 (profiles/r2_notes.md) -- stage A's tile
      // writes at a stride lose more than stage B's contiguous reads gain, so "a" stays the default.
      const char* lay = kn.global_layout.empty() ? nullptr : kn.global_layout.c_str();
      const int tb = sb.strided->fpw;
      int shb = 0;
      while ((1 << shb) < tb) ++shb;
      const long long ngroups_a = n2 / t;
      if (lay != nullptr && lay[0] == 'b' && (1 << shb) == tb && n1 % tb == 0 &&
          (n1 / sa.strided->radices[sa.strided->n_radices - 1]) % tb == 0) {
        // Intermediate laid out per stage-B work-group: block K = k1 / tb holds, for every stage-A group g, the tile
        // [k1 % tb][n2 % t] -- stage B reads its n2 / t tiles as ONE contiguous block (contiguous in / strided out
        // is the fastest shape of these kernels: 5.8 TB/s on C3), stage A writes whole tiles of tb * t elements at
        // a stride of n2 / t tiles.
        const long long tile = static_cast<long long>(tb) * t;
        out.back().sa.out_gdist = tile;                                         // group g starts at tile g of block 0
        out.back().sa.out_tile_shift = shb;                                     // k1 -> (k1 / tb, k1 % tb)
        out.back().sa.out_stride = static_cast<unsigned>(ngroups_a * tile);     // next block K
        out.back().sa.out_tile_mul = static_cast<unsigned>(t);                  // k1 % tb
        out.back().sa.out_fdist = 1;                                            // n2 % t
        sb.sa.in_gdist = ngroups_a * tile;                                      // work-group K reads block K
        sb.sa.in_tile_shift = sh;                                               // n2 -> (n2 / t, n2 % t)
        sb.sa.in_stride = static_cast<unsigned>(tile);
        sb.sa.in_fdist = static_cast<unsigned>(t);
      } else {
        out.back().sa.out_gdist = n1 * t;  // stage A (already pushed)
        out.back().sa.out_stride = static_cast<unsigned>(t);
        out.back().sa.out_fdist = 1;
        sb.sa.in_tile_shift = sh;
        sb.sa.in_stride = static_cast<unsigned>(n1 * t);
        sb.sa.in_fdist = static_cast<unsigned>(t);
      }
      // square tiles: stage B takes its lanes element-fastest inside a tile (strided_pass TIN)
      if (sb.strided->launch_tin != nullptr && sb.strided->fpw == t && !kn.no_tiled_lanes) {
        sb.tiled_in = 1;
      }
      if (split_tiled && sb.strided->fpw == t) sb.tiled_in = 1;  // (mfn_mixed_tin: run_stage)
      // tiles twice as wide as stage B's groups (fp32 n2 = 2048 behind a 16-column stage A)
      if (sb.tiled_in == 0 && sb.strided->launch_tin_w != nullptr && sb.strided->tin_w == t &&
          !kn.no_tiled_lanes && !kn.no_wide_tiles) {
        sb.tiled_in = 2;
      }
      if (ltw && sb.tiled_in == 0) {
        fail(PFFT_INTERNAL_ERROR, "four-step pair: the load-modifier stage B lost its tiled-input form");
      }
      if (ltw) attach_store_tables(sb, n, true);
    }
  }
  // (No tiled intermediate -- n2 or its first pass no multiple of stage A's group width: 10^6 = 1000 x 1000, 68640 = 104 x 660 --
  //  and too long to stage its rows through LDS: stage B reads the row-major intermediate f-fastest, four elements of each of
  //  FPW rows per wave.  A row-lanes read was built and measured in round 5: +5 % for 10^6, -3 ... +2 % on 20 other lengths,
  //  profiles/r5_perf_tin_rows.txt -- the read pattern is not what holds these plans; removed in round 6.)
  if (chunk < count || fs_pair) {  // (a pair's stages may carry their own grid rule / the tiled-input form)
    regrid_for_chunk(out.back(), std::min(chunk, count) * n2);
    regrid_for_chunk(sb, std::min(chunk, count) * n1);
  }
  out.push_back(sb);
  record(PFFT_TIER_GLOBAL, {static_cast<int>(n1), static_cast<int>(n2)}, sb.generic ? GENERIC_WG : kb->wg,
         sb.generic ? sb.ga.fpw : kb->fpw, std::max(sa.lds_bytes, sb.lds_bytes));
  return PFFT_TIER_GLOBAL;
}

}  // namespace pfa
