// plan_t: construction, device tables, kernel look-up and the stage builders every planner uses (plan.hpp).
#include "plan.hpp"

namespace pfa {

namespace {
thread_local hipEvent_t g_armed_stop_event = nullptr;
}
void arm_stop_event(hipEvent_t ev) { g_armed_stop_event = ev; }
hipEvent_t take_stop_event() {
  hipEvent_t ev = g_armed_stop_event;
  g_armed_stop_event = nullptr;
  return ev;
}

std::vector<int> tw_offsets(const std::vector<int>& radices) {
  std::vector<int> off(radices.size(), 0);
  long long ns = 1;
  int o = 0;
  for (size_t p = 0; p < radices.size(); ++p) {
    off[p] = o;
    if (p >= 1) o += static_cast<int>(ns) * (radices[p] - 1);
    ns *= radices[p];
  }
  return off;
}

bool generic_radix_ok(int r) {
  switch (r) {
#define PFA_OK(x) case x:
    PFA_GENERIC_RADICES(PFA_OK)
    PFA_GENERIC_RADICES_BIG(PFA_OK)
#undef PFA_OK
    return true;
    default:
      return false;
  }
}

/// Factorise n into radices the generic kernel implements; largest radices first (fewest LDS passes).
/// Returns an empty vector when n has a prime factor that is not a supported radix.
std::vector<int> choose_radices(long long n) {
  std::vector<int> r;
  if (n == 1) return {1};
  long long rem = n;
  while (rem > 1) {
    int best = 0;
    for (int c = 16; c >= 2; --c) {
      if (rem % c == 0) {
        best = c;
        break;
      }
    }
    if (best == 0) {
      // prime factors up to the wavefront size: what a wave64 build of the reference takes as one sub-group DFT
      // (/root/reference/src/portfft/common/subgroup.hpp:226-253)
      for (int c : {17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61}) {
        if (rem % c == 0) best = c;
      }
    }
    if (best == 0 || !generic_radix_ok(best)) return {};
    // avoid a trailing tiny radix: 16 * 2 -> 8 * 4
    r.push_back(best);
    rem /= best;
  }
  if (r.size() >= 2 && r.back() == 2 && r[r.size() - 2] == 16) {
    r[r.size() - 2] = 8;
    r.back() = 4;
  }
  if (static_cast<int>(r.size()) > GENERIC_MAX_PASSES) return {};
  return r;
}

/// The environment, once per commit (plan.hpp: plan_knobs).  The order of this table is the bit order of `mask`.
plan_knobs plan_knobs::from_env() {
  plan_knobs k;
  int bit = 0;
  auto set = [](const char* name) -> const char* {
    const char* e = getenv(name);
    return e;
  };
  auto mark = [&](bool differs) {
    if (differs) k.mask |= 1ull << bit;
    ++bit;
  };
  auto flag = [&](const char* name, bool* dst) {  // present = the alternative
    *dst = set(name) != nullptr;
    mark(*dst);
  };
  auto onoff = [&](const char* name, bool* dst) {  // default on; "0" switches off
    const char* e = set(name);
    const bool def = *dst;
    if (e != nullptr && e[0] != '\0') *dst = std::atoi(e) != 0;
    mark(*dst != def);
  };
  flag("PFFT_NO_PRECOMPILED", &k.no_precompiled);
  flag("PFFT_XLANE", &k.xlane);
  flag("PFFT_NO_REGRES", &k.no_regres);
  flag("PFFT_NO_LTW", &k.no_ltw);
  flag("PFFT_NO_STW_ROWISH", &k.no_stw_rowish);
  flag("PFFT_JIT_SPEC_RADICES", &k.jit_spec_radices);
  flag("PFFT_NO_MIXED_ROWS", &k.no_mixed_rows);
  flag("PFFT_NO_THREE_STAGE", &k.no_three_stage);
  flag("PFFT_NO_TILED_SCRATCH", &k.no_tiled_scratch);
  flag("PFFT_NO_TILED_LANES", &k.no_tiled_lanes);
  flag("PFFT_TIN_ROWS", &k.tin_rows_gone);  // (no effect any more; keeps the bit order of `mask`)
  flag("PFFT_NO_XCD_LOCAL", &k.no_xcd_local);
  flag("PFFT_ND_TWO_STAGE_COLUMNS", &k.nd_two_stage_columns);
  flag("PFFT_NO_FS_PAIRS", &k.no_fs_pairs);
  flag("PFFT_NO_HALF_PAIRS", &k.no_half_pairs);
  flag("PFFT_NO_SPLIT_RULE", &k.no_split_rule);
  flag("PFFT_NO_SPLIT_TILED", &k.no_split_tiled);
  flag("PFFT_NO_WIDE_TILES", &k.no_wide_tiles);
  onoff("PFFT_SPLIT_CACHED", &k.split_cached);
  onoff("PFFT_PAIR_XCD", &k.pair_xcd);
  onoff("PFFT_STOP_EVENT_ON_LAUNCH", &k.stop_event_on_launch);
  onoff("PFFT_XCD_CHECK", &k.xcd_check);
  if (const char* e = set("PFFT_2D_TWO_PASS")) k.two_pass_2d_off = e[0] == '0';
  mark(k.two_pass_2d_off);
  if (const char* e = set("PFFT_JIT_VERBOSE")) k.jit_verbose = e[0] != '\0' && e[0] != '0';
  mark(k.jit_verbose);
  if (const char* e = set("PFFT_DEBUG_GLOBAL")) {
    k.debug_global_set = true;
    k.debug_global = e;
  }
  mark(k.debug_global_set);
  if (const char* e = set("PFFT_GLOBAL_LAYOUT")) k.global_layout = e;
  mark(!k.global_layout.empty());
  if (const char* e = set("PFFT_GLOBAL_N1")) {
    k.global_n1_set = true;
    k.global_n1 = std::atoll(e);
  }
  mark(k.global_n1_set);
  if (const char* e = set("PFFT_CHUNK_OVERLAP")) {
    if (e[0] != '\0') k.chunk_overlap = std::atoi(e);
  }
  mark(k.chunk_overlap != 2);
  if (const char* e = set("PFFT_JIT_GROUPS_PER_WG")) k.jit_groups_per_wg = std::atoi(e);
  mark(k.jit_groups_per_wg >= 0);
  if (const char* e = set("PFFT_GROUPS_PER_WG")) {
    k.groups_per_wg_set = true;
    k.groups_per_wg = std::atoi(e);
  }
  mark(k.groups_per_wg_set);
  if (const char* e = set("PFFT_GLOBAL_CHUNK_MIB")) {
    k.global_chunk_mib_set = true;
    k.global_chunk_mib = std::atol(e);
  }
  mark(k.global_chunk_mib_set);
  if (const char* e = set("PFFT_CACHE_CHUNK_MIB")) {
    k.cache_chunk_mib_set = true;
    k.cache_chunk_mib = std::atol(e);
  }
  mark(k.cache_chunk_mib_set);
  if (const char* e = set("PFFT_ROW_IN_MAX_N")) k.row_in_max_n = std::atoi(e);
  mark(k.row_in_max_n != 512);
  if (const char* e = set("PFFT_THREE_STAGE_MIN")) k.three_stage_min = std::atoll(e);
  mark(k.three_stage_min > 0);
  if (const char* e = set("PFFT_THREE_STAGE_N3")) k.three_stage_n3 = std::atoll(e);
  mark(k.three_stage_n3 != 0);
  if (const char* e = set("PFFT_XCD_MIN_BATCH")) k.xcd_min_batch = std::atoll(e);
  mark(k.xcd_min_batch >= 0);
  if (const char* e = set("PFFT_XCD_SLOTS")) k.xcd_slots = std::atoi(e);
  mark(k.xcd_slots > 0);
  if (const char* e = set("PFFT_XCD_LAG")) k.xcd_lag = std::atoi(e);
  mark(k.xcd_lag > 0);
  if (const char* e = set("PFFT_XCD_MAX_ITERS")) k.xcd_max_iters = std::atoll(e);
  mark(k.xcd_max_iters >= 0);
  onoff("PFFT_XCD_CONTIG", &k.xcd_contig);
  onoff("PFFT_HX_OVER_REGISTERED", &k.hx_over_registered);
  flag("PFFT_NO_SPLIT_2D_CACHED", &k.no_split_2d_cached);
  if (const char* e = set("PFFT_BI_N1")) k.bi_n1 = std::atoll(e);
  mark(k.bi_n1 > 0);
  flag("PFFT_NO_BI_N1_RULE", &k.no_bi_n1_rule);
  flag("PFFT_NO_BIG_BI", &k.no_big_bi);
  flag("PFFT_NO_BI_WIDE", &k.no_bi_wide);
  flag("PFFT_NO_BI_WIDE_SPLIT2", &k.no_bi_wide_split2);
  if (const char* e = set("PFFT_BI_WIDE_FPW")) k.bi_wide_fpw = std::atoi(e);
  mark(k.bi_wide_fpw > 0);
  flag("PFFT_NO_UNALIGNED_POLICY", &k.no_unaligned_policy);
  flag("PFFT_NO_SPLIT_UNALIGNED_POLICY", &k.no_split_unaligned_policy);
  return k;
}

plan_t::~plan_t() {
  (void)hipStreamSynchronize(stream);
  if (aux_stream != nullptr) {
    (void)hipStreamSynchronize(aux_stream);
    (void)hipStreamDestroy(aux_stream);
  }
  for (hipEvent_t e : chunk_events) (void)hipEventDestroy(e);
  if (scratch != nullptr) (void)hipFree(scratch);
  if (alias_scratch != nullptr) (void)hipFree(alias_scratch);
  if (xcd_ctl != nullptr) (void)hipFree(xcd_ctl);
  if (xcd_tmap != nullptr) (void)hipFree(xcd_tmap);
  if (xcd_report != nullptr) (void)hipHostFree(xcd_report);
}

/// the control block of the XCD-local launch: all zero before its first launch (the kernel keeps it that way)
void plan_t::alloc_xcd_ctl() {
  if (xcd_ctl_bytes == 0 || xcd_ctl != nullptr) return;
  hip_check(hipMalloc(&xcd_ctl, xcd_ctl_bytes), "hipMalloc(control block)");
  // On the plan's own stream and waited for.  hipMemset is not: it returns before the fill has run, the fill sits on
  // the null stream, and a launch on a non-blocking stream does not wait for it -- with more host threads than
  // hardware queues the fill was seen to land in the middle of the plan's first launch
  // (tests/cpp/multi_device_test.cpp with MDT_THREADS=8: counters back at zero, hand-off waits that never end).
  hip_check(hipMemsetAsync(xcd_ctl, 0, xcd_ctl_bytes, stream), "hipMemsetAsync(control block)");
  hip_check(hipMalloc(&xcd_tmap, xcd_tmap_bytes), "hipMalloc(transform records)");
  hip_check(hipMemsetAsync(xcd_tmap, 0, xcd_tmap_bytes, stream), "hipMemsetAsync(transform records)");
  hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
  // the copy's host report: what the recovery launch behind a launch that gave up tells the host (pfft_plan_get_info)
  void* r = nullptr;
  hip_check(hipHostMalloc(&r, XCD_REPORT_WORDS * sizeof(unsigned), hipHostMallocPortable | hipHostMallocMapped),
            "hipHostMalloc(report)");
  std::memset(r, 0, XCD_REPORT_WORDS * sizeof(unsigned));
  xcd_report = static_cast<unsigned*>(r);
}

void* plan_t::upload(const void* host, size_t bytes) {
  void* d = nullptr;
  hip_check(hipMalloc(&d, bytes), "hipMalloc(twiddles)");
  tables->ptrs.push_back(d);
  hip_check(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice), "hipMemcpy(twiddles)");
  twiddle_bytes += bytes;
  return d;
}

void* plan_t::upload_twiddles(const std::vector<int>& radices) {
  if (desc.precision == PFFT_PRECISION_F64) {
    auto t = host_twiddles<double>(radices);
    return upload(t.data(), t.size() * sizeof(double));
  }
  auto t = host_twiddles<float>(radices);
  return upload(t.data(), t.size() * sizeof(float));
}

/// fused N-D kernel: the per-dimension tables one after the other, last dimension first (nd_cfg_type_name)
void* plan_t::upload_nd_twiddles(const nd_kernel& nk) {
  auto build = [&](auto tag) {
    using T = decltype(tag);
    std::vector<T> all;
    for (int d = static_cast<int>(nk.dims.size()) - 1; d >= 0; --d) {
      const std::vector<int>& r = nk.radices[static_cast<size_t>(d)];
      if (r.size() < 2) continue;  // a single pass has no twiddles
      const std::vector<T> t = host_twiddles<T>(r);
      all.insert(all.end(), t.begin(), t.end());
    }
    if (all.empty()) all = {T(1), T(0)};
    return upload(all.data(), all.size() * sizeof(T));
  };
  return desc.precision == PFFT_PRECISION_F64 ? build(double{}) : build(float{});
}

/// W_M^m split in two tables (see generic_args::stw_*)
void plan_t::upload_store_twiddles(long long M, int shift, const void** lo, const void** hi) {
  const long long nlo = 1ll << shift;
  const long long nhi = (M + nlo - 1) / nlo + 1;
  auto fill = [&](auto tag, long long count, long long mult) {
    using T = decltype(tag);
    std::vector<T> v(static_cast<size_t>(2 * count));
    for (long long i = 0; i < count; ++i) {
      const long double a = -2.0L * static_cast<long double>(PI_L) * static_cast<long double>((i * mult) % M) /
                            static_cast<long double>(M);
      v[static_cast<size_t>(2 * i)] = static_cast<T>(cosl(a));
      v[static_cast<size_t>(2 * i + 1)] = static_cast<T>(sinl(a));
    }
    return upload(v.data(), v.size() * sizeof(T));
  };
  if (desc.precision == PFFT_PRECISION_F64) {
    *lo = fill(double{}, nlo, 1);
    *hi = fill(double{}, nhi, nlo);
  } else {
    *lo = fill(float{}, nlo, 1);
    *hi = fill(float{}, nhi, nlo);
  }
}

/// levels / shift of the store-modifier tables of an M-point plan behind kernel k's LDS: the fewest levels (fewest
/// multiplies per root) whose tables stay within 16 KiB and do not cost the kernel a resident work-group; failing
/// that the smallest tables (levels of <= 128 entries).  fp32 N <= 2^20: two levels; fp64 N = 2^20: three.
void plan_t::store_table_shape(const strided_kernel* k, long long M, int* levels, int* shift) const {
  int bits = 0;
  while ((1ll << bits) < M) ++bits;
  bits = std::max(bits, 1);
  const size_t cu_lds = 160 * 1024, own = std::max<size_t>(k->lds_bytes, 1);
  const size_t resident = std::min<size_t>(cu_lds / own, 8);
  for (int l = 1; l <= 4; ++l) {
    const int sh = (bits + l - 1) / l;
    const size_t bytes = (static_cast<size_t>(l) << sh) * elem_bytes();
    if (bytes <= 16 * 1024 && k->lds_bytes + bytes <= max_lds && std::min<size_t>(cu_lds / (own + bytes), 8) >= resident) {
      *levels = l;
      *shift = sh;
      return;
    }
  }
  *levels = std::max(1, (bits + 6) / 7);
  *shift = std::max(1, (bits + *levels - 1) / *levels);
}

/// can stage kernel k carry the tables behind its LDS?  (always, for the kernels the planners produce: their own
/// LDS ends at 144 KiB and the smallest tables take at most 8 KiB)
bool plan_t::store_tables_fit(const strided_kernel* k, long long M) const {
  if (k == nullptr) return false;
  if (k->stw_mode != 1) return true;  // two global tables
  int levels = 0, shift = 0;
  store_table_shape(k, M, &levels, &shift);
  return levels <= 4 && k->lds_bytes + (static_cast<size_t>(levels) << shift) * elem_bytes() <= max_lds;
}

/// Store-modifier tables of a strided stage: L tables of 2^shift entries, table l = W_M^(i << (l * shift)), so that
/// W_M^m is the product of one entry per table (stockham_strided.hpp: stw_from_lds).  The kernel copies them behind
/// its own LDS once per work-group (round 1: two L2-resident tables read with scattered gathers).
/// on_loads: the tables of a stage B that carries the modifier on its loads (strided_kernel::fs_ltw; pre-compiled
/// tiled-input form, tables in LDS): the stage keeps store_modifier == 0, its fn_tin forms get the larger LDS limit.
void plan_t::attach_store_tables(stage& s, long long M, bool on_loads) {
  const strided_kernel* k = s.strided;
  s.store_modifier = on_loads ? 0 : 1;
  if (k->stw_mode != 1 && !on_loads) {  // this kernel's store-modifier forms read two global tables (strided_kernel::stw_mode)
    int sh = 0;
    while ((1ll << (2 * sh)) < M) ++sh;
    upload_store_twiddles(M, sh, &s.sa.stw_lo, &s.sa.stw_hi);
    s.sa.stw_shift = sh;
    return;
  }
  int levels = 0, shift = 0;
  store_table_shape(k, M, &levels, &shift);
  const size_t extra = (static_cast<size_t>(levels) << shift) * elem_bytes();
  s.sa.stw_tab = store_tables_for(M, levels, shift);
  s.sa.stw_levels = levels;
  s.sa.stw_lshift = shift;
  const size_t total = k->lds_bytes + extra;
  finish_store_tables(s, k, total, on_loads);
}

/// device copy of the multi-level tables W_M^(i << (l * shift)), l < levels, i < 2^shift (cached per (M, shift))
const void* plan_t::store_tables_for(long long M, int levels, int shift) {
  auto& slot = store_tables[std::make_pair(M, shift)];
  if (slot == nullptr) {
    const long long per = 1ll << shift;
    auto fill = [&](auto tag) {
      using T = decltype(tag);
      std::vector<T> v(static_cast<size_t>(2 * levels * per));
      for (int l = 0; l < levels; ++l) {
        for (long long i = 0; i < per; ++i) {
          const long long m = static_cast<long long>((static_cast<unsigned long long>(i) << (l * shift)) %
                                                     static_cast<unsigned long long>(M));
          const long double a = -2.0L * static_cast<long double>(PI_L) * static_cast<long double>(m) /
                                static_cast<long double>(M);
          v[static_cast<size_t>(2 * (l * per + i))] = static_cast<T>(cosl(a));
          v[static_cast<size_t>(2 * (l * per + i) + 1)] = static_cast<T>(sinl(a));
        }
      }
      return upload(v.data(), v.size() * sizeof(T));
    };
    slot = desc.precision == PFFT_PRECISION_F64 ? fill(double{}) : fill(float{});
  }
  return slot;
}

void plan_t::finish_store_tables(stage& s, const strided_kernel* k, size_t total, bool on_loads) {
  if (on_loads) {
    for (int d = 0; d < 2; ++d) {
      if (k->fn_tin[d] != nullptr && total > 48 * 1024) {
        hip_check(hipFuncSetAttribute(k->fn_tin[d], hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(total)),
                  "hipFuncSetAttribute");
      }
    }
    s.lds_bytes = total;
    return;
  }
  s.store_modifier = 1;
  if (k->launch != nullptr) {  // pre-compiled: the store-modifier forms get the larger dynamic LDS limit
    for (int d = 0; d < 2; ++d) {
      if (k->fn[d * 2 + 1] != nullptr && total > 48 * 1024) {
        hip_check(hipFuncSetAttribute(k->fn[d * 2 + 1], hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(total)),
                  "hipFuncSetAttribute");
      }
    }
    const long long groups = strided_groups(s.count, s.sa.inner, k->fpw);
    const void* fn = k->fn[s.backward * 2 + 1];
    if (fn != nullptr) s.grid = persistent_grid(fn, nullptr, k->wg, total, groups, k->groups_per_wg);
  }
  s.lds_bytes = total;
}

const spec_kernel* plan_t::find_spec(long long n, bool allow_hx) const {
  if (kn.no_precompiled) return nullptr;  // experiments: planner-chosen kernels everywhere
  int count = 0;
  const spec_kernel* k =
      desc.precision == PFFT_PRECISION_F64 ? spec_kernels_f64(&count) : spec_kernels_f32(&count);
  // PFFT_XLANE: prefer the cross-lane variant of a length (measurement / parity of stockham_xlane.hpp)
  const bool want_xlane = kn.xlane && desc.complex_storage == PFFT_INTERLEAVED_COMPLEX;
  const spec_kernel* found = nullptr;
  const bool no_regres = kn.no_regres || !allow_hx;  // A/B twin of the register-resident entries / forms they do not have
  for (int i = 0; i < count; ++i) {
    if (k[i].n != n || k[i].lds_bytes > max_lds || (k[i].hx != 0 && no_regres)) continue;
    if (k[i].xlane != 0) {
      if (want_xlane) return &k[i];
      continue;
    }
    if (found == nullptr) found = &k[i];
  }
  return found;
}

/// column_both: the stage is column-shaped on both sides -> the wide-group entry of the length, when there is one
/// row_side: one side of the stage is row-shaped -> the row-friendly entry of the length, when there is one
/// fs_stage: 1 / 2 = the length's entry for the four-step stage A / B when there is one (strided_kernel::fs_a / fs_b);
/// entries that exist only for such pairs are invisible to every other request
const strided_kernel* plan_t::find_strided(long long n, bool column_both, bool row_side, long long inner_count,
                                           int policy, bool store_modifier, int fs_stage, bool allow_ltw) const {
  int count = 0;
  const strided_kernel* k =
      desc.precision == PFFT_PRECISION_F64 ? strided_kernels_f64(&count) : strided_kernels_f32(&count);
  const strided_kernel* found = nullptr;
  // (stage B: an entry that carries the modifier on its loads first -- PFFT_NO_LTW=1 hides those entries, their
  //  tiled-input form cannot run without the tables)
  const bool ltw_ok = allow_ltw && !kn.no_ltw;
  for (int pass = 0; pass < 2 && fs_stage != 0; ++pass) {
    for (int i = 0; i < count; ++i) {
      if (k[i].n != n || k[i].lds_bytes > max_lds || k[i].policy != policy) continue;
      if (k[i].fs_ltw != 0 && (!ltw_ok || pass == 1)) continue;
      if (pass == 0 && fs_stage == 2 && k[i].fs_ltw == 0) continue;
      if ((fs_stage == 1 && k[i].fs_a != 0) || (fs_stage == 2 && k[i].fs_b != 0)) return &k[i];
    }
  }
  if (fs_stage != 0) return nullptr;
  // (the narrow pair entries are for stages with one contiguous side: with both sides strided -- batch-interleaved
  //  N = 256 -- 16 columns at four work-groups per CU run at 4.3 TB/s against 5.2 for the 64-column entry)
  for (int i = 0; i < count; ++i) {
    if (k[i].n != n || k[i].lds_bytes > max_lds || k[i].policy != policy) continue;
    if (k[i].fs_only != 0) continue;
    if (k[i].wide == 0 && k[i].rowish == 0 && found == nullptr) found = &k[i];
    // wide groups only pay when the stage has that many adjacent columns (surplus lanes would be masked)
    if (k[i].wide != 0 && column_both && (inner_count < 0 || inner_count >= k[i].fpw)) return &k[i];
    // ... which is also the entry for a stage with the store modifier: its last radix is 8 (fp32 n = 1024: 16.8.8
    // against the 32.32 prefetch kernel, whose radix-32 store butterfly holds 32 modifier values next to 32 outputs:
    // four-step N = 2^20 stage A 147 -> 131 us per 256 MiB, 1.128 -> 1.006 ms per GiB)
    if (k[i].rowish != 0 && (row_side || (store_modifier && !kn.no_stw_rowish)) &&
        k[i].lds_bytes_row <= max_lds) {
      return &k[i];
    }
  }
  return found;
}

/// FFTs per work-group of the strided kernel get_strided(n, inner_count, ...) would deliver; 0 when there is none.
/// Cheap: consults the registry and the runtime planner, compiles nothing.
int plan_t::strided_fpw(long long n, long long inner_count) const {
  const strided_kernel* k = find_strided(n);
  if (k != nullptr) return k->fpw;
  wg_params p;
  if (jit_enabled() && choose_strided_params(desc.precision, n, inner_count, max_lds, &p)) return p.fpw;
  return 0;
}

/// the pre-compiled strided kernel when it suits the stage, otherwise a runtime-specialised one (jit.hpp)
/// policy: cache policy of the stage (strided_kernel::policy; 1 writer -- needs store_modifier --, 2 reader)
const strided_kernel* plan_t::get_strided(long long n, long long inner_count, bool store_modifier, bool user_split,
                                          bool column_both, bool row_side, int policy) {
  if (policy == 3 && jit_enabled() && (!user_split || (column_both && !store_modifier && !row_side))) {  // (row-shaped sides too: the row-staged forms are built from this entry)
    // an unaligned row pitch (aux_of_policy): the kernel compiled at commit on default cache policies with the shared group
    // walk, whatever the registry holds for the length (split user planes: batch-interleaved on both sides only)
    std::string why;
    if (const strided_kernel* k = jit_strided_kernel(desc.precision, n, inner_count, store_modifier, user_split ? 1 : 0, max_lds, &why, column_both, policy)) {
      return k;
    }
    jit_note("strided (unaligned pitch)", n, why);
  }
  if (policy == 3) policy = 0;
  // (split user planes: streamed kernels -- unless a registered policy twin carries its split form: the reader of the
  //  two-pass 2-D plan's second pass, the only caller that asks for a policy on user planes)
  if (user_split && policy != 0) {
    const strided_kernel* t = find_strided(n, column_both, false, inner_count, policy, store_modifier);
    if (t == nullptr || t->launch_split == nullptr) policy = 0;
  }
  const strided_kernel* k = find_strided(n, column_both, row_side && !user_split, inner_count, policy, store_modifier);
  // A registered entry that sits alone on its CU (128 KiB of LDS) against the register-resident form of the same group, two
  // work-groups per CU (stockham_strided_hx.hpp) -- column-shaped stages without the store modifier.  Measured
  // (profiles/r6_hx_over_registered.txt, r6_bi_two_stage_split.txt): fp64 BI N = 1024 x 8 columns 0.578 -> 0.624, at a batch of
  // 66 000 0.543-0.558 -> 0.586; fp32 BI N = 512 x 32 columns (8.8.8 on 1024 lanes) 0.592 / 0.616 -> 0.660 / 0.685.  NOT fp32 n = 1024:
  // its registered 32.32 software-pipelined kernel on 512 lanes holds 0.60-0.64 against 0.54-0.63.  Hence: fp64, and fp32 where
  // the registered entry is one 16-wave work-group.  Costs that length's first commit one hiprtc compilation (then the disk cache).
  if (k != nullptr && kn.hx_over_registered && (desc.precision == PFFT_PRECISION_F64 || (k->wg >= 1024 && k->n <= 512)) &&
      k->lds_bytes > 80 * 1024 &&
      column_both && !store_modifier && !row_side && !user_split && jit_enabled()) {
    wg_params p;
    if (choose_strided_params(desc.precision, n, inner_count, max_lds, &p, column_both, k->fpw) &&
        !strided_hx_candidates(p, max_lds).empty()) {
      std::string why;
      const strided_kernel* j = jit_strided_kernel(desc.precision, n, inner_count, false, 0, max_lds, &why, column_both, policy, k->fpw);
      if (j != nullptr && j->hx != 0) return j;
    }
  }
  if (k != nullptr) return k;
  if (find_strided(n, column_both, row_side && !user_split, inner_count, 0, store_modifier) != nullptr && policy != 0) {
    return find_strided(n, column_both, row_side && !user_split, inner_count, 0, store_modifier);  // no twin registered
  }
  std::string why;
  k = jit_strided_kernel(desc.precision, n, inner_count, store_modifier, user_split ? 1 : 0, max_lds, &why,
                         column_both, policy);
  if (k == nullptr) jit_note("strided", n, why);
  return k;
}

/// four-step stages on SPLIT_COMPLEX data: split user side, interleaved scratch side (runtime-specialised only)
const strided_kernel* plan_t::get_strided_mixed(long long n, long long inner_count, int split_mode, int policy) {
  std::string why;
  return jit_strided_kernel(desc.precision, n, inner_count, split_mode == 2, split_mode, max_lds, &why, false, policy);
}

/// PFFT_JIT_VERBOSE: say why a configuration stayed on the slower tier
void plan_t::jit_note(const char* what, long long n, const std::string& why) const {
  if (kn.jit_verbose && !why.empty()) {
    std::fprintf(stderr, "[portfft_amd jit] %s n=%lld not specialised: %s\n", what, n, why.c_str());
  }
}

/// the pre-compiled packed kernel, otherwise a runtime-specialised one
const spec_kernel* plan_t::get_spec(long long n) {
  std::string why;
  const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
  // 80 ... 152 KiB: where the register-resident planner has a two-work-groups-per-CU plan it goes before a registered or
  // tuned LDS-resident kernel of the length (one work-group per CU) -- tools/perf_hx_pairs.py: fp32 12288 0.49 -> 0.62,
  // 13824 0.52 -> 0.64, 14400 0.48 -> 0.60, 15360 0.52 -> 0.66, fp64 6144 0.56 -> 0.72, 6400 0.52 -> 0.67.  (A plan whose kernel
  // does not fit its register budget hands the length back: jit_spec_kernel.)
  if (!kn.no_regres && jit_enabled() && !kn.jit_spec_radices) {
    wg_params q;
    if (choose_hx_params(desc.precision, n, max_lds, &q) && q.hx_pair != 0) {
      const spec_kernel* reg = find_spec(n);
      if (reg != nullptr && reg->hx != 0) return reg;  // a registered register-resident entry is such a plan already
      if (const spec_kernel* k = jit_spec_kernel(desc.precision, n, split, max_lds, &why, false, nullptr, true); k != nullptr && k->hx != 0) {
        return k;
      }
    }
  }
  if (const spec_kernel* k = find_spec(n)) return k;
  if (plan_measure_enabled() && jit_enabled() && !kn.jit_spec_radices) {
    const std::vector<int> choice = measured_radices(n);
    if (!choice.empty()) {
      if (const spec_kernel* k = jit_spec_kernel(desc.precision, n, split, max_lds, &why, false, &choice)) return k;
    }
  }
  if (jit_enabled() && !kn.jit_spec_radices) {  // the tuned table of this architecture
    const std::vector<int> tuned = builtin_choice(jit_device_arch(), desc.precision, n, false);
    if (!tuned.empty()) {
      if (const spec_kernel* k = jit_spec_kernel(desc.precision, n, split, max_lds, &why, false, &tuned)) return k;
    }
  }
  const spec_kernel* k = jit_spec_kernel(desc.precision, n, split, max_lds, &why, false, nullptr, !kn.no_regres);
  if (k == nullptr) jit_note("packed", n, why);
  return k;
}

/// work-group loop trips of a strided stage (stockham_strided.hpp: strided_ngroups)
long long plan_t::strided_groups(long long count, long long inner, int fpw) {
  return ((count + inner - 1) / inner) * ((inner + fpw - 1) / fpw);
}

/// can the strided kernel `k` address this stage?  (interleaved data, whole groups, 32-bit byte ranges)
bool plan_t::strided_fits(const strided_kernel* k, long long inner_count, int in_buf, const addressing& ia,
                          int out_buf, const addressing& oa) const {
  if (k == nullptr) return false;
  // split storage: both sides user buffers (split variant), both scratch (interleaved variant), or one of each
  // when the entry carries the mixed forms
  const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
  if (split && ((in_buf == BUF_SCRATCH) != (out_buf == BUF_SCRATCH))) {
    if (k->mfn_mixed[in_buf == BUF_SCRATCH ? 2 : 0] == nullptr) return false;
  }
  (void)inner_count;
  auto range_ok = [&](const addressing& a) {
    const unsigned long long elems = static_cast<unsigned long long>(k->fpw - 1) * a.dist_inner +
                                     static_cast<unsigned long long>(k->n - 1) * a.stride + 1;
    return a.stride < (1ll << 31) && a.dist_inner < (1ll << 31) && elems * elem_bytes() < 0xFFFFFFF0ull;
  };
  return range_ok(ia) && range_ok(oa);
}

/// W_n^m for m in [0, n): the inter-pass column twiddles of the two-pass 2-D plan
const void* plan_t::upload_unit_roots(long long n) {
  auto fill = [&](auto tag) {
    using T = decltype(tag);
    std::vector<T> v(static_cast<size_t>(2 * n));
    for (long long i = 0; i < n; ++i) {
      const long double a = -2.0L * static_cast<long double>(PI_L) * static_cast<long double>(i) /
                            static_cast<long double>(n);
      v[static_cast<size_t>(2 * i)] = static_cast<T>(cosl(a));
      v[static_cast<size_t>(2 * i + 1)] = static_cast<T>(sinl(a));
    }
    return upload(v.data(), v.size() * sizeof(T));
  };
  return desc.precision == PFFT_PRECISION_F64 ? fill(double{}) : fill(float{});
}

stage plan_t::make_strided_stage(const strided_kernel* k, long long count, long long inner_count, int in_buf,
                                 const addressing& ia, int out_buf, const addressing& oa, double scale, int backward,
                                 int store_modifier, bool allow_row) {
  stage s;
  s.strided = k;
  s.store_modifier = store_modifier;
  s.n = k->n;
  s.in_buf = in_buf;
  s.out_buf = out_buf;
  s.count = count;
  s.in_addr = ia;
  s.out_addr = oa;
  s.backward = backward;
  strided_args& a = s.sa;
  a.tw = upload_twiddles(std::vector<int>(k->radices, k->radices + k->n_radices));
  a.total = count;
  a.inner = std::max<long long>(inner_count, 1);
  a.in_dist_outer = ia.dist_outer;
  a.out_dist_outer = oa.dist_outer;
  a.in_stride = static_cast<unsigned>(ia.stride);
  a.out_stride = static_cast<unsigned>(oa.stride);
  a.in_fdist = static_cast<unsigned>(ia.dist_inner);
  a.out_fdist = static_cast<unsigned>(oa.dist_inner);
  a.scale = scale;
  a.stw_tab = nullptr;
  a.stw_levels = 0;
  a.stw_lshift = 0;
  a.stw_cdiv = 1;
  a.stw_lo = nullptr;
  a.stw_hi = nullptr;
  a.stw_shift = 0;
  s.lds_bytes = k->lds_bytes;
  // row-shaped side of an interleaved fp32 stage: copy it through LDS with full-line accesses
  const bool user_split =
      desc.complex_storage == PFFT_SPLIT_COMPLEX && in_buf != BUF_SCRATCH && out_buf != BUF_SCRATCH;
  // fp32 row-shaped sides are staged through LDS (`_row` forms).  Measured (tools/perf_global_f32.py): a staged
  // row-shaped INPUT pays up to n = 512 (four-step stage B of N=65536: 2.6 vs 2.0 TB/s; P->BI n=256 5.2 vs 3.2) and
  // loses beyond (N=2^20: 1.76 vs 2.07 with the group-major intermediate; P->BI n=1024 2.7 vs 3.7); a staged
  // row-shaped OUTPUT always pays (BI->P n=1024 4.0 vs 2.0).
  int want_row = 0;
  if (ia.stride == 1 && ia.dist_inner != 1 && oa.dist_inner == 1 && (k->n <= kn.row_in_max_n || k->rowish != 0)) want_row = 1;
  if (oa.stride == 1 && oa.dist_inner != 1 && ia.dist_inner == 1) want_row = 2;
  if (!allow_row) want_row = 0;  // four-step pair: stage B reads the group-major intermediate (tiled-input form)
  const bool mixed = desc.complex_storage == PFFT_SPLIT_COMPLEX && (in_buf == BUF_SCRATCH) != (out_buf == BUF_SCRATCH);
  if (k->launch == nullptr && want_row != 0 && !user_split && !mixed) {  // runtime-compiled entry: build the row form
    std::string why;
    if (jit_strided_ensure_row(k, want_row - 1, max_lds, &why)) {
      s.row_mode = want_row;
      s.lds_bytes = k->lds_bytes_row;
    }
  }
  // mixed stage B (interleaved scratch rows -> the user's planes): always row-staged when the image fits -- its
  // f-fastest form reads 8 bytes per lane from FPW different rows, and there is no tiled-input form to fall back on
  // (power-of-two rows only: fp32 N = 65536 1.88 -> 0.91 ms per GiB, 2^20 1.61 -> 1.28; with 8000-byte rows
  //  -- N = 10^6 -- the f-fastest form spreads over the channels by itself and the staged form loses, 1.84 -> 2.21)
  if (allow_row && k->launch == nullptr && mixed && in_buf == BUF_SCRATCH && ia.stride == 1 && ia.dist_inner != 1 &&
      oa.dist_inner == 1 && (k->n & (k->n - 1)) == 0 && !kn.no_mixed_rows) {
    std::string why;
    if (jit_strided_ensure_row(k, 0, max_lds, &why, 3)) {
      s.row_mode = 1;
      s.lds_bytes = k->lds_bytes_row;
    }
  }
  if (k->launch_row != nullptr && !user_split && k->lds_bytes_row <= max_lds &&
      (want_row == 0 || k->fn_row[(want_row - 1) * 2 + backward] != nullptr)) {  // pre-compiled entries
    s.row_mode = want_row;
    if (s.row_mode != 0) {
      s.lds_bytes = k->lds_bytes_row;
      for (int i = 0; i < 4; ++i) {
        if (k->fn_row[i] == nullptr) continue;  // policy twins carry the row-shaped-input forms only
        hip_check(hipFuncSetAttribute(k->fn_row[i], hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(k->lds_bytes_row)),
                  "hipFuncSetAttribute");
      }
    }
  }
  for (int i = 0; i < 4 && k->launch != nullptr; ++i) {
    if (k->lds_bytes > 48 * 1024) {
      if (k->fn[i] != nullptr) {
        hip_check(hipFuncSetAttribute(k->fn[i], hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(k->lds_bytes)),
                  "hipFuncSetAttribute");
      }
      if (k->fn_split[i / 2] != nullptr) {
        hip_check(hipFuncSetAttribute(k->fn_split[i / 2], hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(k->lds_bytes)),
                  "hipFuncSetAttribute");
      }
      if (k->fn_tin_w[i / 2] != nullptr) {
        hip_check(hipFuncSetAttribute(k->fn_tin_w[i / 2], hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(k->lds_bytes)),
                  "hipFuncSetAttribute");
      }
      if (k->fn_tin[i / 2] != nullptr) {
        hip_check(hipFuncSetAttribute(k->fn_tin[i / 2], hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(k->lds_bytes)),
                  "hipFuncSetAttribute");
      }
    }
  }
  const long long groups = strided_groups(count, a.inner, k->fpw);
  if (k->launch == nullptr && s.row_mode != 0) {
    s.grid = persistent_grid(nullptr, mixed ? k->mfn_row_mixed[backward] : k->mfn_row[(s.row_mode - 1) * 2 + backward],
                             k->wg, k->lds_bytes_row, groups, 1);
  } else if (k->launch == nullptr) {  // runtime-compiled: whichever variant this stage will launch
    const bool split_storage = desc.complex_storage == PFFT_SPLIT_COMPLEX;
    hipFunction_t f = user_split ? (store_modifier ? k->mfn_split_stw[backward] : k->mfn_split[backward]) : k->mfn[backward * 2];
    if (split_storage && (in_buf == BUF_SCRATCH) != (out_buf == BUF_SCRATCH)) {
      f = k->mfn_mixed[(in_buf == BUF_SCRATCH ? 2 : 0) + backward];
    }
    if (f == nullptr) f = k->mfn[backward * 2 + 1];
    s.grid = persistent_grid(nullptr, f, k->wg, k->lds_bytes, groups, k->groups_per_wg);
  } else if (s.row_mode != 0) {
    s.grid = persistent_grid(k->fn_row[(s.row_mode - 1) * 2 + backward], nullptr, k->wg, k->lds_bytes_row, groups,
                             k->groups_per_wg);
  } else {
    const void* fn = k->fn[backward * 2 + (store_modifier ? 1 : 0)];
    if (fn == nullptr) fn = k->fn[backward * 2] != nullptr ? k->fn[backward * 2] : k->fn[backward * 2 + 1];
    s.grid = persistent_grid(fn, nullptr, k->wg, k->lds_bytes, groups, k->groups_per_wg);
  }
  return s;
}

/// the launch grid of a chunked stage is sized for ONE chunk (`count` FFTs / `nmat` matrices), not for the whole
/// batch: the per-kernel rule "groups_per_wg groups per work-group" must hold inside a chunk
void plan_t::regrid_for_chunk(stage& s, long long count) {
  if (s.strided != nullptr) {
    const strided_kernel* k = s.strided;
    const long long groups = strided_groups(count, s.sa.inner, k->fpw);
    if (k->launch == nullptr) {  // runtime-compiled entries: one group per work-group unless asked otherwise
      int gpw = s.gpw;
      if (kn.jit_groups_per_wg >= 0) gpw = kn.jit_groups_per_wg;  // experiments
      if (gpw > 1 && s.row_mode == 0) {
        hipFunction_t f = nullptr;
        for (hipFunction_t c : {k->mfn[0], k->mfn[1], k->mfn[2], k->mfn[3], k->mfn_mixed[0], k->mfn_mixed[2]}) {
          if (f == nullptr) f = c;
        }
        if (f != nullptr) s.grid = persistent_grid(nullptr, f, k->wg, std::max(k->lds_bytes, s.lds_bytes), groups, gpw);
      }
      return;
    }
    const void* fn = s.row_mode != 0 ? k->fn_row[(s.row_mode - 1) * 2 + s.backward]
                                     : (s.tiled_in == 2 ? k->fn_tin_w[s.backward]
                                        : s.tiled_in != 0 ? k->fn_tin[s.backward] : k->fn[s.backward * 2 + (s.store_modifier ? 1 : 0)]);
    if (fn == nullptr) return;
    const size_t lds = s.row_mode != 0 ? k->lds_bytes_row : std::max(k->lds_bytes, s.lds_bytes);
    s.grid = persistent_grid(fn, nullptr, k->wg, lds, groups, s.gpw > 0 ? s.gpw : k->groups_per_wg);
  } else if (s.rows2d != nullptr) {
    const rows2d_kernel* k = s.rows2d;
    const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
    s.grid = persistent_grid(k->launch != nullptr ? (split ? k->fn_split : k->fn)[s.backward] : nullptr,
                             k->mfn[s.backward], k->wg, k->lds_bytes,
                             count / std::max<long long>(1, s.ra.n0) * (s.ra.n0 / k->rc), k->groups_per_wg);
  }
}

/// `count` transforms in chunks of AT MOST `chunk` (the cap is the caller's: the scratch allocation, the Infinity Cache):
/// the same number of chunks, equally filled.  N = 40000 x 3355 in 256 MiB chunks was 4 chunks of 838 transforms + one of 3
/// -- two launches of an almost empty grid per execute; now five chunks of 671.  (Round 4 dropped a sliver chunk and let
/// the others grow past the cap by up to an eighth: 288 MiB "cache-sized" chunks fall off the Infinity Cache -- ADVICE r4.)
long long plan_t::even_chunks(long long chunk, long long count) {
  if (chunk >= count) return count;
  const long long n = (count + chunk - 1) / chunk;
  return (count + n - 1) / n;
}

/// bytes of intermediate data per chunk of the GLOBAL tier = cap of the scratch allocation
/// (PFFT_GLOBAL_CHUNK_MIB overrides; 0 = unbounded)
size_t plan_t::global_chunk_bytes() const {
  if (kn.global_chunk_mib_set) {
    if (kn.global_chunk_mib <= 0) return ~size_t{0} >> 1;
    return static_cast<size_t>(kn.global_chunk_mib) << 20;
  }
  return size_t{4} << 30;
}

/// Two-launch plans (four-step tier, two-pass 2-D plan) run chunk by chunk with the intermediate of a chunk sized
/// to the 256 MiB Infinity Cache: the first launch writes it with default-policy stores (streamed loads), the second
/// reads it with default-policy loads (streamed stores), so the intermediate's read is served on-die.  Measured on
/// C5 (tools/tune_2d_small.hip): 256 matrices as 8 chunks of 256 MiB 1.274 ms against 1.421 ms unchunked with
/// streamed accesses; chunks of 288 MiB and more fall off the cliff (1.60 ms), smaller ones pay launch tails.
/// PFFT_CACHE_CHUNK_MIB overrides (0: no cache-sized chunks, everything streamed as in round 1).
size_t plan_t::cache_chunk_bytes() const {
  if (kn.cache_chunk_mib_set) return kn.cache_chunk_mib <= 0 ? 0 : static_cast<size_t>(kn.cache_chunk_mib) << 20;
  return size_t{256} << 20;
}

/// Grid of a persistent kernel.  Measured on the N=4096 kernel (tools/probes/proto_c2.hip, interleaved rounds): a grid of
/// 1-2x the resident work-groups keeps every work-group in lock-step (all load, then all compute) and loses ~5 %
/// against a grid where each work-group handles only `groups_per_wg` groups (4-5 is the optimum when the kernel
/// pre-loads its twiddles into registers, 1 when it re-reads them per FFT): staggered work-group start times smooth
/// the HBM demand.
unsigned plan_t::persistent_grid(const void* fn, hipFunction_t mfn, int wg, size_t lds, long long groups,
                                 int groups_per_wg) {
  int per_cu = 0;
  if (fn != nullptr) {
    hip_check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, wg, lds), "occupancy query");
  } else {
    hip_check(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mfn, wg, lds), "occupancy query");
  }
  per_cu = std::max(per_cu, 1);
  if (kn.groups_per_wg_set) groups_per_wg = kn.groups_per_wg;  // grid-rule experiments
  const long long resident = static_cast<long long>(per_cu) * n_cus;
  // groups_per_wg comes from the per-kernel tuning (tools/tune.hip, profiles/r1_notes.md); 0 selects the long
  // persistent loop, which only the one-work-group-per-CU kernels (f32 N=16384) prefer
  long long grid = groups_per_wg <= 0 ? 2 * resident : (groups + groups_per_wg - 1) / groups_per_wg;
  grid = std::min(groups, std::max(grid, std::min<long long>(groups, 2 * resident)));
  grid = std::min<long long>(grid, 1ll << 30);
  return static_cast<unsigned>(std::max<long long>(1, grid));
}

stage plan_t::make_spec_stage(const spec_kernel* k, long long count, int in_buf, long long in_off, int out_buf,
                              long long out_off, double scale, int backward, const void* twiddles,
                              const unpacked_kernel* unpacked) {
  stage s;
  s.generic = false;
  s.spec = k;
  s.n = k->n;
  s.in_buf = in_buf;
  s.out_buf = out_buf;
  s.in_offset = in_off;
  s.out_offset = out_off;
  s.count = count;
  s.scale = scale;
  s.backward = backward;
  s.tw = twiddles != nullptr ? twiddles
                             : upload_twiddles(std::vector<int>(k->radices, k->radices + k->n_radices));
  for (int d = 0; d < 2 && k->launch != nullptr; ++d) {
    if (k->lds_bytes > 48 * 1024) {
      hip_check(hipFuncSetAttribute(k->fn[d], hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(k->lds_bytes)),
                "hipFuncSetAttribute");
      hip_check(hipFuncSetAttribute(k->fn_split[d], hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(k->lds_bytes)),
                "hipFuncSetAttribute");
    }
  }
  const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
  s.unpacked = unpacked;
  if (unpacked != nullptr) {
    s.grid = persistent_grid(nullptr, (split ? unpacked->fn_split : unpacked->fn)[backward], k->wg, k->lds_bytes,
                             (count + k->fpw - 1) / k->fpw, k->groups_per_wg);
    return s;
  }
  s.grid = persistent_grid(k->launch != nullptr ? k->fn[backward] : nullptr,
                           split ? k->mfn_split[backward] : k->mfn[backward], k->wg, k->lds_bytes,
                           (count + k->fpw - 1) / k->fpw, k->groups_per_wg);
  return s;
}

stage plan_t::make_generic_stage(long long n, long long count, long long inner_count, int in_buf, const addressing& ia,
                                 int out_buf, const addressing& oa, double scale, int conj_in, int conj_out) {
  const std::vector<int> radices = choose_radices(n);
  if (radices.empty()) {
    fail(PFFT_UNSUPPORTED_CONFIGURATION, "FFT size ", n, " : Large Prime sized FFT currently is unsupported");
  }
  stage s;
  s.generic = true;
  s.n = static_cast<int>(n);
  s.in_buf = in_buf;
  s.out_buf = out_buf;
  s.count = count;
  s.in_addr = ia;
  s.out_addr = oa;
  generic_args& g = s.ga;
  g.n = static_cast<int>(n);
  g.n_passes = (n == 1) ? 0 : static_cast<int>(radices.size());
  const std::vector<int> offs = tw_offsets(radices);
  for (int p = 0; p < g.n_passes; ++p) {
    g.radix[p] = radices[static_cast<size_t>(p)];
    g.tw_off[p] = offs[static_cast<size_t>(p)];
  }
  g.tw = upload_twiddles(radices);
  g.in_stride = ia.stride;
  g.out_stride = oa.stride;
  g.in_dist_inner = ia.dist_inner;
  g.in_dist_outer = ia.dist_outer;
  g.out_dist_inner = oa.dist_inner;
  g.out_dist_outer = oa.dist_outer;
  g.inner_count = std::max<long long>(inner_count, 1);
  g.total_count = count;
  g.conj_in = conj_in;
  g.conj_out = conj_out;
  g.scale = scale;
  g.stw_lo = nullptr;
  g.stw_hi = nullptr;
  g.stw_shift = 0;
  // FFTs per work-group: fill ~64 KiB of LDS (two images), at least one FFT, at most what the stage has
  const size_t per_fft = 2 * static_cast<size_t>(n) * elem_bytes();
  long long fpw = std::max<long long>(1, static_cast<long long>((64 * 1024) / per_fft));
  fpw = std::min<long long>(fpw, std::max<long long>(1, count));
  fpw = std::min<long long>(fpw, 256);
  g.fpw = static_cast<int>(fpw);
  s.lds_bytes = per_fft * static_cast<size_t>(fpw);
  if (s.lds_bytes > max_lds) {
    fail(PFFT_OUT_OF_LOCAL_MEMORY, "FFT size ", n, " needs ", s.lds_bytes, " bytes of LDS, device has ", max_lds);
  }
  // lane order: walk whichever index is contiguous in memory
  g.magic_n = generic_magic(static_cast<unsigned>(n));
  g.magic_fpw = generic_magic(static_cast<unsigned>(fpw));
  {
    unsigned ns = 1;
    for (int p = 0; p < g.n_passes; ++p) {
      g.magic_nb[p] = generic_magic(static_cast<unsigned>(n / g.radix[p]));
      g.magic_ns[p] = generic_magic(ns);
      ns *= static_cast<unsigned>(g.radix[p]);
    }
  }
  g.in_f_fast = (fpw > 1 && ia.dist_inner < ia.stride) ? 1 : 0;
  g.out_f_fast = (fpw > 1 && oa.dist_inner < oa.stride) ? 1 : 0;
  bool big_radix = false;  // a prime radix 37 ... 61: the kernel's "big radix" instantiation (generic_args.hpp)
  for (int p = 0; p < g.n_passes; ++p) big_radix = big_radix || g.radix[p] > GENERIC_MAX_SMALL_RADIX;
  const void* fn = generic_kernel_symbol(desc.precision, big_radix);
  if (s.lds_bytes > 48 * 1024) {
    hip_check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(max_lds)),
              "hipFuncSetAttribute");
  }
  s.grid = persistent_grid(fn, nullptr, GENERIC_WG, s.lds_bytes, (count + fpw - 1) / fpw, 1);
  return s;
}

/// `forced_n1`: the first factor of the four-step split (measured_split's candidates; 0: the planner's rules)
plan_t::plan_t(const pfft_desc_t& d, hipStream_t s, long long forced_n1_) : desc(d), stream(s), forced_n1(forced_n1_) {
  validate(desc);
  hip_check(hipGetDevice(&device), "hipGetDevice");
  hipDeviceProp_t prop;
  hip_check(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
  n_cus = prop.multiProcessorCount;
  max_lds = prop.sharedMemPerBlock;
  info.rank = desc.rank;
  info.n_compute_units = n_cus;
  info.knob_mask = kn.mask;
  build_direction(PFFT_FORWARD);
  build_direction(PFFT_BACKWARD);
  if (scratch_bytes > 0) {
    hip_check(hipMalloc(&scratch, scratch_bytes), "hipMalloc(scratch)");
  }
  alloc_xcd_ctl();
  if (alias_scratch_bytes > 0) ensure_alias_scratch();
  info.twiddle_bytes = twiddle_bytes;
  info.scratch_bytes = scratch_bytes + alias_scratch_bytes;
  for (int d = 0; d < 2; ++d) {
    long long n = 0;
    for (const stage& st : stages[d]) {
      const long long batches = st.chunk_group < 0 ? 1 : st.count / std::max<long long>(1, st.ffts_per_batch);
      const long long per = std::max<long long>(1, st.chunk_batches);
      n += st.chunk_group < 0 ? 1 : (batches + per - 1) / per;
      if (st.xcd != nullptr) ++n;  // its recovery launch (two when the execute's buffers alias)
    }
    info.launches[d] = static_cast<int32_t>(std::min<long long>(n, 0x7fffffff));
  }
}

/// Copy of a committed plan (committed_descriptor_impl.hpp:774-817): the kernels and the twiddle tables are shared,
/// the scratch buffers are allocated again, so two copies can execute concurrently on two streams.
plan_t::plan_t(const plan_t& o) : kn(o.kn), desc(o.desc), stream(o.stream), device(o.device), n_cus(o.n_cus), max_lds(o.max_lds),
               tables(o.tables), scratch_bytes(o.scratch_bytes), twiddle_bytes(o.twiddle_bytes),
               alias_scratch_bytes(o.alias_scratch_bytes), two_pass_chunk_bytes(o.two_pass_chunk_bytes),
               n_chunk_groups(o.n_chunk_groups), info(o.info), overlap_scratch_half(o.overlap_scratch_half),
               overlap_mode(o.overlap_mode) {
  stages[0] = o.stages[0];
  stages[1] = o.stages[1];
  xcd_ctl_bytes = o.xcd_ctl_bytes;
  xcd_tmap_bytes = o.xcd_tmap_bytes;
  device_guard dg(device);
  if (scratch_bytes > 0) hip_check(hipMalloc(&scratch, scratch_bytes), "hipMalloc(scratch)");
  alloc_xcd_ctl();
  if (o.alias_scratch != nullptr) ensure_alias_scratch();
}

}  // namespace pfa
