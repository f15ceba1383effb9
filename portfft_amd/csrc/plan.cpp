// Planner + executor behind pfft_plan_create / pfft_execute (the committed_descriptor of the reference).
//
// Mirrors the role of /root/reference/src/portfft/committed_descriptor_impl.hpp:
//   ctor + prepare_implementation (:210-313, :716-768)      -> plan_t::plan_t / plan_1d
//   calculate_twiddles (:430-434, per-level dispatchers)     -> twiddle_table (host long double -> device)
//   allocate_scratch_and_precompute_scan (:579-708)          -> scratch for the GLOBAL tier
//   dispatch_direction / dispatch_dimensions (:852-950)      -> build_direction: the stage list of one direction
//   run_kernel (:1088-1111)                                  -> run_stage
// The structure is our own: a plan is two flat lists of kernel launches ("stages"), one per direction, resolved at
// commit time; execute only binds the user pointers and enqueues them on the plan's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <memory>
#include <vector>

#include "descriptor.hpp"
#include "jit.hpp"
#include "kernels.hpp"

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

namespace {

void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) fail(PFFT_HIP_ERROR, what, ": ", hipGetErrorString(e));
}

enum buffer_id { BUF_IN = 0, BUF_OUT = 1, BUF_SCRATCH = 2 };

/// Addressing of the FFTs of one stage, in complex elements relative to the stage's buffer.
/// FFT t starts at offset + (t / inner_count) * dist_outer + (t % inner_count) * dist_inner.
struct addressing {
  long long offset = 0;
  long long stride = 1;
  long long dist_inner = 0;
  long long dist_outer = 0;
};

struct stage {
  bool generic = false;
  const spec_kernel* spec = nullptr;
  const unpacked_kernel* unpacked = nullptr;  // spec + UNPACKED layout: in_addr / out_addr hold strides and distances
  const strided_kernel* strided = nullptr;
  strided_args sa{};
  const rows2d_kernel* rows2d = nullptr;  // first pass of the two-pass 2-D plan (stockham_rows2d.hpp)
  rows2d_args ra{};
  const xcd_kernel* xcd = nullptr;  // XCD-local four-step launch (stockham_xcd.hpp): both stages of N = n1 x n2
  xcd_args xa{};
  unsigned recover_grid = 0;  // work-groups of the recovery launch behind it (stockham_xcd_recover_kernel)
  // two-pass 2-D plan: pass 1 permutes rows between distinct buffers (IN -> OUT), pass 2 works in place on OUT.
  // When the caller's buffers alias (in-place transform) the intermediate goes through scratch instead:
  // 1: this stage writes it (out_buf -> scratch), 2: this stage reads it (in_buf -> scratch)
  int alias_scratch = 0;
  int store_modifier = 0;
  int row_mode = 0;  // 0: both sides addressed by the passes, 1: row-shaped input staged, 2: row-shaped output staged
  int tiled_in = 0;  // 1: the kernel's tiled-input form (strided_kernel::launch_tin)
  int gpw = 0;       // > 0: groups per work-group of this stage instead of the kernel's own rule (four-step pairs)
  int in_buf = BUF_IN, out_buf = BUF_OUT;
  long long count = 0;  // number of FFTs
  unsigned grid = 1;
  // spec
  long long in_offset = 0, out_offset = 0;
  const void* tw = nullptr;
  double scale = 1.0;
  int backward = 0;
  // generic
  generic_args ga{};
  addressing in_addr, out_addr;
  size_t lds_bytes = 0;
  // info
  int n = 0;
  // GLOBAL tier: the stages of one transform run chunk by chunk so that the intermediate stays cache resident
  int chunk_group = -1;            // stages with the same id advance together
  long long chunk_batches = 0;     // user transforms (2-D plan: matrices) per chunk of this group
  long long ffts_per_batch = 0;    // FFTs this stage runs per user transform
  long long in_batch_dist = 0;     // elements between consecutive user transforms in the stage's input (0: scratch)
  long long out_batch_dist = 0;
};

const double PI_L = 3.14159265358979323846264338327950288;

template <typename T>
std::vector<T> host_twiddles(const std::vector<int>& radices) {
  // table layout shared by every kernel: pass p >= 1 with stride Ns = r0*...*r(p-1) owns (r_p - 1) * Ns entries,
  // entry [(t-1)*Ns + q] = W_{Ns*r_p}^{t*q}.  Computed in long double, rounded once.
  std::vector<T> tw;
  long long ns = 1;
  for (size_t p = 0; p < radices.size(); ++p) {
    const int r = radices[p];
    if (p >= 1) {
      for (int t = 1; t < r; ++t) {
        for (long long q = 0; q < ns; ++q) {
          const long double a = -2.0L * static_cast<long double>(PI_L) * static_cast<long double>(t * q) /
                                static_cast<long double>(ns * r);
          tw.push_back(static_cast<T>(cosl(a)));
          tw.push_back(static_cast<T>(sinl(a)));
        }
      }
    }
    ns *= r;
  }
  if (tw.empty()) {
    tw.push_back(T(1));
    tw.push_back(T(0));
  }
  return tw;
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

}  // namespace

/// Device allocations that copies of a plan share (twiddle tables): freed when the last copy goes away.
/// (reference: the kernels and twiddles of committed_descriptor_impl are shared_ptr members, copied by
/// create_copy, committed_descriptor_impl.hpp:774-803)
struct shared_allocs {
  std::vector<void*> ptrs;
  ~shared_allocs() {
    for (void* p : ptrs) (void)hipFree(p);
  }
};

/// makes the plan's device current for the duration of a call when it is not (plan_t::device was unused)
struct device_guard {
  int prev = -1;
  explicit device_guard(int device) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != device) {
      if (hipSetDevice(device) != hipSuccess) fail(PFFT_HIP_ERROR, "hipSetDevice(", device, ") failed");
      prev = cur;
    }
  }
  ~device_guard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  device_guard(const device_guard&) = delete;
  device_guard& operator=(const device_guard&) = delete;
};

/// Per-launch state of the chunk-overlap paths.  It travels as an argument of run_stage, never as plan state: a
/// launch that throws cannot leave a later execute() with a barrier-free launch or a shifted scratch base.
struct launch_ctx {
  bool on_aux = false;       // launch on aux_stream
  bool any_order = false;    // launch without the in-order barrier (strided / rows2d stages)
  size_t scratch_shift = 0;  // bytes added to the scratch base (the half of a double-buffered scratch in use)
};

struct plan_t {
  pfft_desc_t desc{};
  hipStream_t stream = nullptr;
  int device = 0;
  int n_cus = 0;
  size_t max_lds = 0;
  long long forced_n1 = 0;
  std::vector<stage> stages[2];
  std::shared_ptr<shared_allocs> tables = std::make_shared<shared_allocs>();  // twiddles: shared by copies
  void* scratch = nullptr;                                                     // scratch: one per copy
  size_t scratch_bytes = 0;
  size_t twiddle_bytes = 0;
  void* xcd_ctl = nullptr;        // control block of the XCD-local four-step launch (xcd_args.hpp): one per copy
  size_t xcd_ctl_bytes = 0;
  void* xcd_tmap = nullptr;       // ... its per-transform records (xcd_args::tmap), one per copy
  size_t xcd_tmap_bytes = 0;
  unsigned* xcd_report = nullptr;  // ... and the copy's host report (pinned; XCD_REPORT_WORDS words, xcd_args.hpp)
  unsigned xcd_recoveries_seen = 0;  // PFFT_XCD_CHECK=1: report[0] at the last check
  const bool xcd_check = xcd_check_enabled();  // fixed at commit
  void* alias_scratch = nullptr;  // intermediate of the two-pass 2-D plan for aliasing (in-place) executes
  size_t alias_scratch_bytes = 0;
  size_t two_pass_chunk_bytes = 0;  // bytes of one chunk of the two-pass 2-D plan (what an aliasing execute needs)
  int tail_policy = 0;              // cache policy plan_1d gives the strided stage it plans (two-pass 2-D plan: reader)
  std::map<std::pair<long long, int>, const void*> store_tables;  // attach_store_tables: (M, shift) -> device tables
  int n_chunk_groups = 0;
  pfft_plan_info_t info{};
  // chunk overlap (execute): the second launch of chunk c runs on aux_stream while the first launch of chunk c + 1
  // runs on the plan's stream, so the tail of one fills with the head of the other
  hipStream_t aux_stream = nullptr;
  std::vector<hipEvent_t> chunk_events;
  size_t overlap_scratch_half = 0;  // bytes of one half when the chunks of a four-step plan double-buffer the scratch

  /// How consecutive chunks of a two-launch plan overlap (PFFT_CHUNK_OVERLAP):
  /// 0 not at all: every launch in order on the plan's stream;
  /// 1 second launches on a second stream behind events (measured 20 % slower: profiles/r2_notes.md);
  /// 2 (default) the first launch of chunk c + 1 is enqueued without the in-order barrier (hipExtAnyOrderLaunch), so
  ///   it fills the tail of the second launch of chunk c.  A runtime that ignores the flag runs them in order.
  static int chunk_overlap_mode() {
    const char* e = getenv("PFFT_CHUNK_OVERLAP");
    if (e == nullptr || e[0] == '\0') return 2;  // measured: C3 (with cache-sized chunks) +3.7 %, C5 / ref65536 +1.2 %
    return std::atoi(e);
  }
  const int overlap_mode = chunk_overlap_mode();  // fixed at commit
  bool chunk_overlap_enabled() const { return overlap_mode != 0; }


  int scalar_bytes() const { return desc.precision == PFFT_PRECISION_F64 ? 8 : 4; }
  size_t elem_bytes() const { return 2 * static_cast<size_t>(scalar_bytes()); }

  ~plan_t() {
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
  void alloc_xcd_ctl() {
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

  void* upload(const void* host, size_t bytes) {
    void* d = nullptr;
    hip_check(hipMalloc(&d, bytes), "hipMalloc(twiddles)");
    tables->ptrs.push_back(d);
    hip_check(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice), "hipMemcpy(twiddles)");
    twiddle_bytes += bytes;
    return d;
  }

  void* upload_twiddles(const std::vector<int>& radices) {
    if (desc.precision == PFFT_PRECISION_F64) {
      auto t = host_twiddles<double>(radices);
      return upload(t.data(), t.size() * sizeof(double));
    }
    auto t = host_twiddles<float>(radices);
    return upload(t.data(), t.size() * sizeof(float));
  }

  /// fused N-D kernel: the per-dimension tables one after the other, last dimension first (nd_cfg_type_name)
  void* upload_nd_twiddles(const nd_kernel& nk) {
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
  void upload_store_twiddles(long long M, int shift, const void** lo, const void** hi) {
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
  void store_table_shape(const strided_kernel* k, long long M, int* levels, int* shift) const {
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
  bool store_tables_fit(const strided_kernel* k, long long M) const {
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
  void attach_store_tables(stage& s, long long M, bool on_loads = false) {
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
  const void* store_tables_for(long long M, int levels, int shift) {
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

  void finish_store_tables(stage& s, const strided_kernel* k, size_t total, bool on_loads) {
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

  /// Width of the intermediate's tiles -- i.e. the group width its stage A must have -- when `fb` is the four-step
  /// stage B of length n2: its own group width (square tiles, launch_tin) or, `wide`, twice that (launch_tin_w).
  /// 0 when the entry has no such form or the length does not divide into those tiles.
  static int pair_tile(const strided_kernel* fb, long long n2, bool wide) {
    const int t = wide ? (fb->launch_tin_w != nullptr ? fb->tin_w : 0) : (fb->launch_tin != nullptr ? fb->fpw : 0);
    if (t <= 0 || (t & (t - 1)) != 0 || n2 % t != 0 || (n2 / fb->radices[0]) % t != 0) return 0;
    return t;
  }

  const spec_kernel* find_spec(long long n) const {
    if (getenv("PFFT_NO_PRECOMPILED") != nullptr) return nullptr;  // experiments: planner-chosen kernels everywhere
    int count = 0;
    const spec_kernel* k =
        desc.precision == PFFT_PRECISION_F64 ? spec_kernels_f64(&count) : spec_kernels_f32(&count);
    // PFFT_XLANE: prefer the cross-lane variant of a length (measurement / parity of stockham_xlane.hpp)
    const bool want_xlane = getenv("PFFT_XLANE") != nullptr && desc.complex_storage == PFFT_INTERLEAVED_COMPLEX;
    const spec_kernel* found = nullptr;
    const bool no_regres = getenv("PFFT_NO_REGRES") != nullptr;  // A/B twin of the register-resident entries
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
  const strided_kernel* find_strided(long long n, bool column_both = false, bool row_side = false,
                                     long long inner_count = -1, int policy = 0, bool store_modifier = false,
                                     int fs_stage = 0, bool allow_ltw = true) const {
    int count = 0;
    const strided_kernel* k =
        desc.precision == PFFT_PRECISION_F64 ? strided_kernels_f64(&count) : strided_kernels_f32(&count);
    const strided_kernel* found = nullptr;
    // (stage B: an entry that carries the modifier on its loads first -- PFFT_NO_LTW=1 hides those entries, their
    //  tiled-input form cannot run without the tables)
    const bool ltw_ok = allow_ltw && getenv("PFFT_NO_LTW") == nullptr;
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
      if (k[i].rowish != 0 && (row_side || (store_modifier && getenv("PFFT_NO_STW_ROWISH") == nullptr)) &&
          k[i].lds_bytes_row <= max_lds) {
        return &k[i];
      }
    }
    return found;
  }

  /// FFTs per work-group of the strided kernel get_strided(n, inner_count, ...) would deliver; 0 when there is none.
  /// Cheap: consults the registry and the runtime planner, compiles nothing.
  int strided_fpw(long long n, long long inner_count) const {
    const strided_kernel* k = find_strided(n);
    if (k != nullptr) return k->fpw;
    wg_params p;
    if (jit_enabled() && choose_strided_params(desc.precision, n, inner_count, max_lds, &p)) return p.fpw;
    return 0;
  }

  /// the pre-compiled strided kernel when it suits the stage, otherwise a runtime-specialised one (jit.hpp)
  /// policy: cache policy of the stage (strided_kernel::policy; 1 writer -- needs store_modifier --, 2 reader)
  const strided_kernel* get_strided(long long n, long long inner_count, bool store_modifier, bool user_split,
                                    bool column_both = false, bool row_side = false, int policy = 0) {
    if (user_split) policy = 0;
    const strided_kernel* k = find_strided(n, column_both, row_side && !user_split, inner_count, policy, store_modifier);
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
  const strided_kernel* get_strided_mixed(long long n, long long inner_count, int split_mode, int policy = 0) {
    std::string why;
    return jit_strided_kernel(desc.precision, n, inner_count, split_mode == 2, split_mode, max_lds, &why, false, policy);
  }

  /// PFFT_JIT_VERBOSE: say why a configuration stayed on the slower tier
  static void jit_note(const char* what, long long n, const std::string& why) {
    const char* e = getenv("PFFT_JIT_VERBOSE");
    if (e != nullptr && e[0] != '\0' && e[0] != '0' && !why.empty()) {
      std::fprintf(stderr, "[portfft_amd jit] %s n=%lld not specialised: %s\n", what, n, why.c_str());
    }
  }

  /// the pre-compiled packed kernel, otherwise a runtime-specialised one
  const spec_kernel* get_spec(long long n) {
    if (const spec_kernel* k = find_spec(n)) return k;
    std::string why;
    const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
    if (plan_measure_enabled() && jit_enabled() && getenv("PFFT_JIT_SPEC_RADICES") == nullptr) {
      const std::vector<int> choice = measured_radices(n);
      if (!choice.empty()) {
        if (const spec_kernel* k = jit_spec_kernel(desc.precision, n, split, max_lds, &why, false, &choice)) return k;
      }
    }
    if (jit_enabled() && getenv("PFFT_JIT_SPEC_RADICES") == nullptr) {  // the tuned table of this architecture
      const std::vector<int> tuned = builtin_choice(jit_device_arch(), desc.precision, n, false);
      if (!tuned.empty()) {
        if (const spec_kernel* k = jit_spec_kernel(desc.precision, n, split, max_lds, &why, false, &tuned)) return k;
      }
    }
    const spec_kernel* k = jit_spec_kernel(desc.precision, n, split, max_lds, &why);
    if (k == nullptr) jit_note("packed", n, why);
    return k;
  }

  /// Measured planning of the four-step split (PFFT_PLAN_MEASURE=1): every n1 x n2 with both factors in 32 ... 4096, no
  /// more than 16 : 1 apart, that the strided tier can run -- the planner's own choice first -- is committed as a plan of its own (this descriptor, a batch
  /// of 256 MiB, plan_t's forced_n1), timed forward on the plan's stream, and the winner is recorded next to the code
  /// objects like the radix choices (`choice_<arch>_<f32|f64>_<n>.txt` holds "n1 n2").
  long long measured_split(long long n, long long count, long long static_n1) {
    const std::string arch = jit_device_arch();
    const std::vector<int> rec = plan_choice_lookup(arch, desc.precision, n, 1 << 20);
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
      if (getenv("PFFT_JIT_VERBOSE") != nullptr) {
        std::fprintf(stderr, "[portfft_amd plan] n=%lld split %lld x %lld %.3f ms per %lld transforms%s\n", n, c, n / c, ms / 5,
                     batch, ok ? "" : " (failed)");
      }
      if (ok && ms < best_ms) {
        best_ms = ms;
        best = c;
      }
    }
    if (best_ms < 1e30) plan_choice_store(arch, desc.precision, n, {static_cast<int>(best), static_cast<int>(n / best)});
    return best;
  }

  /// buffers and events of a measurement at commit
  struct measure_scratch {
    void *in = nullptr, *out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool alloc(size_t bytes) { return hipMalloc(&in, bytes) == hipSuccess && hipMalloc(&out, bytes) == hipSuccess; }
    measure_scratch() = default;
    measure_scratch(const measure_scratch&) = delete;
    measure_scratch& operator=(const measure_scratch&) = delete;
    ~measure_scratch() {
      if (e0 != nullptr) (void)hipEventDestroy(e0);
      if (e1 != nullptr) (void)hipEventDestroy(e1);
      if (in != nullptr) (void)hipFree(in);
      if (out != nullptr) (void)hipFree(out);
    }
  };

  /// uniform(-1, 1) scalars: a 1 MiB host block replicated by doubling copies on the plan's stream
  void fill_uniform(void* dst, size_t bytes) {
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
  std::vector<int> measured_radices(long long n) {
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
      void* tw = nullptr;
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
      (void)hipFree(tw);
      if (getenv("PFFT_JIT_VERBOSE") != nullptr) {
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

  /// work-group loop trips of a strided stage (stockham_strided.hpp: strided_ngroups)
  static long long strided_groups(long long count, long long inner, int fpw) {
    return ((count + inner - 1) / inner) * ((inner + fpw - 1) / fpw);
  }

  /// can the strided kernel `k` address this stage?  (interleaved data, whole groups, 32-bit byte ranges)
  bool strided_fits(const strided_kernel* k, long long inner_count, int in_buf, const addressing& ia, int out_buf,
                    const addressing& oa) const {
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

  const rows2d_kernel* find_rows2d(long long n1, long long n0, int policy, bool split = false) {
    const char* e = getenv("PFFT_2D_TWO_PASS");
    if (e != nullptr && e[0] == '0') return nullptr;  // experiments / parity A-B: rows, then full-length columns
    int count = 0;
    const rows2d_kernel* k = rows2d_kernels(&count);
    for (int i = 0; i < count; ++i) {
      if (k[i].precision == desc.precision && k[i].n == n1 && k[i].lds_bytes <= max_lds && n0 % k[i].rc == 0 &&
          n0 / k[i].rc >= 2 && k[i].policy == policy && (!split || k[i].launch_split != nullptr)) {
        return &k[i];
      }
    }
    // other row lengths / column counts: the same template instantiated at commit (jit.cpp) -- when the full-length
    // column pass it replaces would move segments below 256 bytes (measured, tools/perf_2d.py: 1080 x 1920 +27 %,
    // 1536^2 +29 %, 3000 x 1000 2.6x, 4096^2 2.1x; with 256-byte column segments available, 384^2 ... 960^2, the
    // runtime-planned first pass does not beat rows + columns)
    const int col_fpw = strided_fpw(n0, n1);
    if (col_fpw > 0 && static_cast<size_t>(col_fpw) * elem_bytes() >= 256) return nullptr;
    // column radices whose remaining n0 / rc points one full-width column pass can take (plan_1d would otherwise
    // answer with two column stages through scratch, and the two-pass plan would be dropped: 3000 x 1000 with rc 2);
    // failing that, any radix with a column kernel at all
    const int full_fpw = desc.precision == PFFT_PRECISION_F64 ? 8 : 16;
    int wide_mask = 0, any_mask = 0;
    for (int rc : {8, 4, 2}) {
      if (n0 % rc != 0 || n0 / rc < 2) continue;
      const int fpw = strided_fpw(n0 / rc, rc * n1);
      if (fpw >= full_fpw) wide_mask |= rc;
      if (fpw > 0) any_mask |= rc;
    }
    const int rc_mask = wide_mask != 0 ? wide_mask : any_mask;
    if (rc_mask == 0) return nullptr;
    std::string why;
    const rows2d_kernel* jk = jit_rows2d_kernel(desc.precision, n1, n0, max_lds, &why, policy, split ? 1 : 0, rc_mask);
    if (jk == nullptr) jit_note("rows2d", n1, why);
    return jk;
  }

  /// W_n^m for m in [0, n): the inter-pass column twiddles of the two-pass 2-D plan
  const void* upload_unit_roots(long long n) {
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

  stage make_rows2d_stage(const rows2d_kernel* k, long long nmat, long long n0, long long in_off, long long out_off,
                          int backward) {
    stage s;
    s.rows2d = k;
    s.n = k->n;
    s.in_buf = BUF_IN;
    s.out_buf = BUF_OUT;
    s.in_offset = in_off;
    s.out_offset = out_off;
    s.count = nmat * n0;
    s.backward = backward;
    s.lds_bytes = k->lds_bytes;
    s.alias_scratch = 1;
    s.ra.tw = upload_twiddles(std::vector<int>(k->radices, k->radices + k->n_radices));
    s.ra.twc = upload_unit_roots(n0);
    s.ra.nmat = nmat;
    s.ra.n0 = static_cast<int>(n0);
    const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
    for (int d = 0; d < 2 && k->launch != nullptr; ++d) {
      if (k->lds_bytes > 48 * 1024) {
        hip_check(hipFuncSetAttribute((split ? k->fn_split : k->fn)[d], hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(k->lds_bytes)),
                  "hipFuncSetAttribute");
      }
    }
    s.grid = persistent_grid(k->launch != nullptr ? (split ? k->fn_split : k->fn)[backward] : nullptr, k->mfn[backward],
                             k->wg, k->lds_bytes, nmat * (n0 / k->rc), k->groups_per_wg);
    return s;
  }

  stage make_strided_stage(const strided_kernel* k, long long count, long long inner_count, int in_buf,
                           const addressing& ia, int out_buf, const addressing& oa, double scale, int backward,
                           int store_modifier = 0, bool allow_row = true) {
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
    if (ia.stride == 1 && ia.dist_inner != 1 && oa.dist_inner == 1 && (k->n <= 512 || k->rowish != 0)) want_row = 1;
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
        oa.dist_inner == 1 && (k->n & (k->n - 1)) == 0 && getenv("PFFT_NO_MIXED_ROWS") == nullptr) {
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
  void regrid_for_chunk(stage& s, long long count) {
    if (s.strided != nullptr) {
      const strided_kernel* k = s.strided;
      const long long groups = strided_groups(count, s.sa.inner, k->fpw);
      if (k->launch == nullptr) {  // runtime-compiled entries: one group per work-group unless asked otherwise
        int gpw = s.gpw;
        if (const char* e = getenv("PFFT_JIT_GROUPS_PER_WG")) gpw = std::atoi(e);  // experiments
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

  /// `count` transforms in chunks of at most `chunk`: the same number of chunks, equally filled -- and one chunk fewer when
  /// the last one would be a sliver (under an eighth of a chunk; the others grow by that much).  N = 40000 x 3355 in 256 MiB
  /// chunks was 4 chunks of 838 transforms + one of 3: two launches of an almost empty grid per execute.
  static long long even_chunks(long long chunk, long long count) {
    if (chunk >= count) return count;
    long long n = (count + chunk - 1) / chunk;
    if (n > 1 && count % chunk != 0 && count % chunk < chunk / 8) --n;
    return (count + n - 1) / n;
  }

  /// bytes of intermediate data per chunk of the GLOBAL tier = cap of the scratch allocation
  /// (PFFT_GLOBAL_CHUNK_MIB overrides; 0 = unbounded)
  static size_t global_chunk_bytes() {
    if (const char* e = getenv("PFFT_GLOBAL_CHUNK_MIB")) {
      const long v = std::atol(e);
      if (v <= 0) return ~size_t{0} >> 1;
      return static_cast<size_t>(v) << 20;
    }
    return size_t{4} << 30;
  }

  /// Two-launch plans (four-step tier, two-pass 2-D plan) run chunk by chunk with the intermediate of a chunk sized
  /// to the 256 MiB Infinity Cache: the first launch writes it with default-policy stores (streamed loads), the second
  /// reads it with default-policy loads (streamed stores), so the intermediate's read is served on-die.  Measured on
  /// C5 (tools/tune_2d_small.hip): 256 matrices as 8 chunks of 256 MiB 1.274 ms against 1.421 ms unchunked with
  /// streamed accesses; chunks of 288 MiB and more fall off the cliff (1.60 ms), smaller ones pay launch tails.
  /// PFFT_CACHE_CHUNK_MIB overrides (0: no cache-sized chunks, everything streamed as in round 1).
  static size_t cache_chunk_bytes() {
    if (const char* e = getenv("PFFT_CACHE_CHUNK_MIB")) {
      const long v = std::atol(e);
      return v <= 0 ? 0 : static_cast<size_t>(v) << 20;
    }
    return size_t{256} << 20;
  }

  /// largest length the generic tier can hold (two LDS images)
  long long generic_max_n() const { return static_cast<long long>(max_lds / (2 * elem_bytes())); }

  /// Grid of a persistent kernel.  Measured on the N=4096 kernel (tools/probes/proto_c2.hip, interleaved rounds): a grid of
  /// 1-2x the resident work-groups keeps every work-group in lock-step (all load, then all compute) and loses ~5 %
  /// against a grid where each work-group handles only `groups_per_wg` groups (4-5 is the optimum when the kernel
  /// pre-loads its twiddles into registers, 1 when it re-reads them per FFT): staggered work-group start times smooth
  /// the HBM demand.
  unsigned persistent_grid(const void* fn, hipFunction_t mfn, int wg, size_t lds, long long groups,
                           int groups_per_wg) {
    int per_cu = 0;
    if (fn != nullptr) {
      hip_check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, wg, lds), "occupancy query");
    } else {
      hip_check(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mfn, wg, lds), "occupancy query");
    }
    per_cu = std::max(per_cu, 1);
    if (const char* e = getenv("PFFT_GROUPS_PER_WG")) groups_per_wg = std::atoi(e);  // grid-rule experiments
    const long long resident = static_cast<long long>(per_cu) * n_cus;
    // groups_per_wg comes from the per-kernel tuning (tools/tune.hip, profiles/r1_notes.md); 0 selects the long
    // persistent loop, which only the one-work-group-per-CU kernels (f32 N=16384) prefer
    long long grid = groups_per_wg <= 0 ? 2 * resident : (groups + groups_per_wg - 1) / groups_per_wg;
    grid = std::min(groups, std::max(grid, std::min<long long>(groups, 2 * resident)));
    grid = std::min<long long>(grid, 1ll << 30);
    return static_cast<unsigned>(std::max<long long>(1, grid));
  }

  stage make_spec_stage(const spec_kernel* k, long long count, int in_buf, long long in_off, int out_buf,
                        long long out_off, double scale, int backward, const void* twiddles = nullptr,
                        const unpacked_kernel* unpacked = nullptr) {
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

  stage make_generic_stage(long long n, long long count, long long inner_count, int in_buf, const addressing& ia,
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

  /// BATCH_INTERLEAVED on both sides (element i of transform b at i * B + b), length n = n1 * n2, B transforms --
  /// `outer` such arrays n * B elements apart (the long column dimension of an N-D array: outer matrices, B adjacent
  /// columns; outer = 1 for a batch-interleaved 1-D descriptor).
  /// Stage A: for every (c, b): FFT over r of x[(r*n2 + c)*B + b], times W_n^{k1*c}, into scratch (same layout).
  /// Stage B: for every (k1, b): FFT over c of scratch[(k1*n2 + c)*B + b] -> out[(k2*n1 + k1)*B + b].
  /// Only taken when a single work-group would hold fewer than 16 (fp32) / 8 (fp64) columns of the whole length.
  bool plan_batch_interleaved_two_stage(std::vector<stage>& out, long long n, long long B, long long outer, int in_buf,
                                        int out_buf, const addressing& ia, const addressing& oa, double scale,
                                        int backward, pfft_dim_info_t* info) {
    const int full_fpw = desc.precision == PFFT_PRECISION_F64 ? 8 : 16;
    if (strided_fpw(n, B) >= full_fpw) return false;
    if (static_cast<unsigned long long>(n) * static_cast<unsigned long long>(B) * elem_bytes() >= 0xFFFFFFF0ull) {
      return false;  // a stage's byte offsets must fit the 32-bit buffer addressing
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
    const long long n2 = n / n1;
    // the intermediate is written once and read once: keep it in the Infinity Cache when all of it fits
    // (measured with random data, tools/perf_cache.py: +4...13 % from 128 MiB of intermediate up; below that the
    //  streamed kernels are faster -- everything sits in the cache anyway -- so small problems keep them)
    const bool cached = cache_chunk_bytes() > 0 && need <= cache_chunk_bytes() && need >= cache_chunk_bytes() / 2;
    const strided_kernel* ka = get_strided(n1, n2 * B, true, false, true, false, cached ? 1 : 0);  // column-shaped on
    const strided_kernel* kb = get_strided(n2, B, false, false, true, false, cached ? 2 : 0);      // both sides: wide
    addressing a_in{ia.offset, n2 * B, 1, n * B};
    addressing a_out{0, n2 * B, 1, n * B};
    addressing b_in{0, B, 1, n2 * B};
    addressing b_out{oa.offset, n1 * B, 1, B};
    if (!strided_fits(ka, n2 * B, in_buf, a_in, BUF_SCRATCH, a_out) || !store_tables_fit(ka, n) ||
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
  bool plan_three_stage(std::vector<stage>& out, long long n, long long count, const addressing& ia,
                        const addressing& oa, double scale, int backward, pfft_dim_info_t* info) {
    if (getenv("PFFT_NO_THREE_STAGE") != nullptr || getenv("PFFT_NO_PRECOMPILED") != nullptr ||
        getenv("PFFT_DEBUG_GLOBAL") != nullptr || getenv("PFFT_NO_TILED_SCRATCH") != nullptr ||
        getenv("PFFT_NO_TILED_LANES") != nullptr) {
      return false;
    }
    long long min_n = (1ll << 22) + 1;  // beyond 2048 x 2048 a two-factor split needs n > 2048 (5 * 2^20: 0.121 against 0.222)
    if (const char* e = getenv("PFFT_THREE_STAGE_MIN")) min_n = std::atoll(e);  // experiments
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
    const long long want_n3 = getenv("PFFT_THREE_STAGE_N3") != nullptr ? std::atoll(getenv("PFFT_THREE_STAGE_N3")) : 0;
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
  int xcd_queue_count() {
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
  bool plan_xcd_local(std::vector<stage>& out, long long n, long long count, const addressing& ia, const addressing& oa,
                      double scale, int backward, pfft_dim_info_t* info) {
    if (desc.complex_storage != PFFT_INTERLEAVED_COMPLEX || getenv("PFFT_NO_XCD_LOCAL") != nullptr ||
        getenv("PFFT_NO_PRECOMPILED") != nullptr || getenv("PFFT_GLOBAL_N1") != nullptr ||
        getenv("PFFT_DEBUG_GLOBAL") != nullptr) {
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
    if (const char* e = getenv("PFFT_XCD_MIN_BATCH")) min_batch = std::atoll(e);
    if (count < min_batch) return false;
    const long long n1 = k->n1, n2 = k->n2;
    const int t = k->fpw;
    int tsh = 0;
    while ((1 << tsh) < t) ++tsh;
    int slots = k->slots, lag = k->lag, lookahead = k->lookahead;
    if (const char* e = getenv("PFFT_XCD_SLOTS")) slots = std::atoi(e);  // schedule experiments
    if (const char* e = getenv("PFFT_XCD_LAG")) lag = std::atoi(e);
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
    if (const char* e = getenv("PFFT_XCD_MAX_ITERS")) x.max_iters = static_cast<unsigned>(std::atoll(e));  // tests: provoke a failed launch
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

  int plan_1d(std::vector<stage>& out, long long n, long long count, long long inner_count, int in_buf,
              const addressing& ia, int out_buf, const addressing& oa, bool packed_io, double scale, int backward,
              pfft_dim_info_t* info) {
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
        const spec_kernel* k = find_spec(n);
        if (k != nullptr && k->hx != 0) k = nullptr;  // (no UNPACKED form of the register-resident entries)
        if (k == nullptr) {
          std::string why;
          k = jit_spec_kernel(desc.precision, n, !interleaved, max_lds, &why, true);
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
    // Long batch-interleaved transforms: one work-group could hold only a few columns (narrow HBM segments), so
    // split N = n1 * n2 and run both four-step stages column shaped with full-width groups, through scratch.
    if (interleaved && desc.rank == 1 && in_buf == BUF_IN && out_buf == BUF_OUT && ia.dist_inner == 1 &&
        oa.dist_inner == 1 && ia.stride == count && oa.stride == count && inner_count == count) {
      if (plan_batch_interleaved_two_stage(out, n, count, 1, in_buf, out_buf, ia, oa, scale, backward, info)) {
        return PFFT_TIER_GLOBAL;
      }
    }
    // ... and long column dimensions of N-D arrays: `inner_count` adjacent columns per array, arrays n * inner apart
    if (interleaved && desc.rank > 1 && in_buf != BUF_SCRATCH && out_buf != BUF_SCRATCH && ia.dist_inner == 1 &&
        oa.dist_inner == 1 && ia.stride == inner_count && oa.stride == inner_count && inner_count > 0 &&
        count % inner_count == 0 && ia.dist_outer == n * inner_count && oa.dist_outer == n * inner_count &&
        getenv("PFFT_ND_TWO_STAGE_COLUMNS") == nullptr) {
      if (plan_batch_interleaved_two_stage(out, n, inner_count, count / inner_count, in_buf, out_buf, ia, oa, scale,
                                           backward, info)) {
        return PFFT_TIER_GLOBAL;
      }
    }
    // the strided tier pays when at least one side is "column" shaped (consecutive FFTs adjacent in memory)
    const bool column_shaped = ia.dist_inner == 1 || oa.dist_inner == 1;
    const bool user_split = !interleaved && in_buf != BUF_SCRATCH;
    const bool column_both = ia.dist_inner == 1 && oa.dist_inner == 1;
    const bool row_side = (ia.stride == 1 && ia.dist_inner != 1) || (oa.stride == 1 && oa.dist_inner != 1);
    if (const strided_kernel* k =
            column_shaped ? get_strided(n, inner_count, false, user_split, column_both, row_side, tail_policy) : nullptr;
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
    if (const char* e = getenv("PFFT_GLOBAL_N1")) want_n1 = std::atoll(e);  // experiments: force the first factor of the split
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
        getenv("PFFT_NO_FS_PAIRS") == nullptr && getenv("PFFT_NO_HALF_PAIRS") == nullptr &&
        getenv("PFFT_NO_TILED_SCRATCH") == nullptr && getenv("PFFT_NO_TILED_LANES") == nullptr &&
        getenv("PFFT_NO_PRECOMPILED") == nullptr && getenv("PFFT_DEBUG_GLOBAL") == nullptr) {
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
        jit_enabled() && want_n1 == 0 && getenv("PFFT_NO_SPLIT_RULE") == nullptr &&
        getenv("PFFT_DEBUG_GLOBAL") == nullptr) {
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
    if (want_n1 == 0 && jit_enabled() && getenv("PFFT_DEBUG_GLOBAL") == nullptr &&
        getenv("PFFT_NO_SPLIT_RULE") == nullptr) {  // the tuned table of this architecture (pairs included: it is measured)
      const std::vector<int> tuned = builtin_choice(jit_device_arch(), desc.precision, n, true);
      if (tuned.size() == 2 && strided_fpw(tuned[0], tuned[1]) > 0 && strided_fpw(tuned[1], tuned[0]) > 0) {
        n1 = tuned[0];
        n2 = tuned[1];
      }
    }
    if (want_n1 == 0 && plan_measure_enabled() && jit_enabled() && in_buf == BUF_IN && out_buf == BUF_OUT &&
        getenv("PFFT_DEBUG_GLOBAL") == nullptr) {
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
    const bool split_cached = getenv("PFFT_SPLIT_CACHED") == nullptr || std::atoi(getenv("PFFT_SPLIT_CACHED")) != 0;
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
    if (interleaved_user && getenv("PFFT_NO_FS_PAIRS") == nullptr && getenv("PFFT_NO_TILED_SCRATCH") == nullptr &&
        getenv("PFFT_NO_TILED_LANES") == nullptr && getenv("PFFT_NO_PRECOMPILED") == nullptr) {
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
    if (!fs_pair && interleaved_user && jit_enabled() && getenv("PFFT_NO_FS_PAIRS") == nullptr &&
        getenv("PFFT_NO_HALF_PAIRS") == nullptr && getenv("PFFT_NO_TILED_SCRATCH") == nullptr &&
        getenv("PFFT_NO_TILED_LANES") == nullptr && getenv("PFFT_NO_PRECOMPILED") == nullptr &&
        getenv("PFFT_DEBUG_GLOBAL") == nullptr && find_strided(n1, false, false, -1, 0, true, 1) == nullptr) {
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
               (half_pair || ka->fn[0] != nullptr) && getenv("PFFT_NO_LTW") == nullptr &&
               getenv("PFFT_DEBUG_GLOBAL") == nullptr &&
               strided_fits(ka, n2, in_buf, addressing{ia.offset, n2, 1, n}, BUF_SCRATCH, addressing{0, n2, 1, n});
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
    const char* dbg = getenv("PFFT_DEBUG_GLOBAL");  // debugging aid: "ga" / "gb" force the generic kernel for a stage
    const bool force_generic_a = dbg != nullptr && std::strstr(dbg, "ga") != nullptr;
    const bool force_generic_b = dbg != nullptr && std::strstr(dbg, "gb") != nullptr;
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
        sa.strided == ka && getenv("PFFT_NO_SPLIT_TILED") == nullptr && getenv("PFFT_NO_TILED_SCRATCH") == nullptr &&
        getenv("PFFT_NO_TILED_LANES") == nullptr) {
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
    if (sa.strided != nullptr && sb.strided != nullptr && sb.row_mode == 0 && getenv("PFFT_NO_TILED_SCRATCH") == nullptr) {
      const int t = sa.strided->fpw;
      int sh = 0;
      while ((1 << sh) < t) ++sh;
      const long long nb0 = n2 / sb.strided->radices[0];
      if ((1 << sh) == t && n2 % t == 0 && nb0 % t == 0 &&
          static_cast<unsigned long long>(n) * elem_bytes() < 0xFFFFFFF0ull) {
        // PFFT_GLOBAL_LAYOUT=b (experiment): intermediate contiguous per stage-B work-group instead of per stage-A
        // work-group.  Measured on C3: 1.675-1.702 ms against 1.660-1.668 (profiles/r2_notes.md) -- stage A's tile
        // writes at a stride lose more than stage B's contiguous reads gain, so "a" stays the default.
        const char* lay = getenv("PFFT_GLOBAL_LAYOUT");
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
        if (sb.strided->launch_tin != nullptr && sb.strided->fpw == t && getenv("PFFT_NO_TILED_LANES") == nullptr) {
          sb.tiled_in = 1;
        }
        if (split_tiled && sb.strided->fpw == t) sb.tiled_in = 1;  // (mfn_mixed_tin: run_stage)
        // tiles twice as wide as stage B's groups (fp32 n2 = 2048 behind a 16-column stage A)
        if (sb.tiled_in == 0 && sb.strided->launch_tin_w != nullptr && sb.strided->tin_w == t &&
            getenv("PFFT_NO_TILED_LANES") == nullptr && getenv("PFFT_NO_WIDE_TILES") == nullptr) {
          sb.tiled_in = 2;
        }
        if (ltw && sb.tiled_in == 0) {
          fail(PFFT_INTERNAL_ERROR, "four-step pair: the load-modifier stage B lost its tiled-input form");
        }
        if (ltw) attach_store_tables(sb, n, true);
      }
    }
    if (chunk < count || fs_pair) {  // (a pair's stages may carry their own grid rule / the tiled-input form)
      regrid_for_chunk(out.back(), std::min(chunk, count) * n2);
      regrid_for_chunk(sb, std::min(chunk, count) * n1);
    }
    out.push_back(sb);
    record(PFFT_TIER_GLOBAL, {static_cast<int>(n1), static_cast<int>(n2)}, sb.generic ? GENERIC_WG : kb->wg,
           sb.generic ? sb.ga.fpw : kb->fpw, std::max(sa.lds_bytes, sb.lds_bytes));
    return PFFT_TIER_GLOBAL;
  }

  void build_direction(int direction) {
    std::vector<stage>& st = stages[direction];
    const int inv = direction == PFFT_FORWARD ? PFFT_BACKWARD : PFFT_FORWARD;
    const view_t vin = view_of(desc, direction), vout = view_of(desc, inv);
    const bool packed = layout_of(desc, direction) == PFFT_LAYOUT_PACKED && layout_of(desc, inv) == PFFT_LAYOUT_PACKED;
    const double scale = direction == PFFT_FORWARD ? desc.forward_scale : desc.backward_scale;
    const int backward = direction == PFFT_BACKWARD ? 1 : 0;
    const long long B = static_cast<long long>(desc.number_of_transforms);
    const int rank = desc.rank;
    const long long total = static_cast<long long>(flattened_length(desc));
    const bool record = direction == PFFT_FORWARD;
    if (rank == 1) {
      const long long n = static_cast<long long>(desc.lengths[0]);
      addressing ia{static_cast<long long>(vin.offset), static_cast<long long>(vin.strides[0]),
                    static_cast<long long>(vin.distance), 0};
      addressing oa{static_cast<long long>(vout.offset), static_cast<long long>(vout.strides[0]),
                    static_cast<long long>(vout.distance), 0};
      plan_1d(st, n, B, B, BUF_IN, ia, BUF_OUT, oa, packed, scale, backward, record ? &info.dims[0] : nullptr);
      return;
    }
    // N-D, packed (validated): contiguous dimension first, then every outer dimension in place on the output,
    // as strided FFTs (reference: dispatch_dimensions, committed_descriptor_impl.hpp:923-948; there one launch per
    // (batch, outer index), here one launch per dimension).
    // ... unless a suffix of the dimensions fits LDS: one fused launch (stockham_nd.hpp) does lengths[s..rank) for
    // every index of the dimensions before it -- the whole transform when s == 0 -- and only lengths[0..s) remain
    // as strided passes
    int fused_from = rank;  // first dimension covered by the fused stage
    for (int s0 = 0; s0 + 2 <= rank && fused_from == rank; ++s0) {
      const std::vector<long long> dims(desc.lengths + s0, desc.lengths + rank);
      std::string why;
      const nd_kernel* nk =
          jit_nd_kernel(desc.precision, dims, desc.complex_storage == PFFT_SPLIT_COMPLEX, max_lds, &why);
      if (nk == nullptr) continue;
      long long outer = 1;
      for (int i = 0; i < s0; ++i) outer *= static_cast<long long>(desc.lengths[i]);
      st.push_back(make_spec_stage(&nk->k, B * outer, BUF_IN, static_cast<long long>(vin.offset), BUF_OUT,
                                   static_cast<long long>(vout.offset), scale, backward, upload_nd_twiddles(*nk)));
      for (int i = s0; record && i < rank; ++i) {
        pfft_dim_info_t& di = info.dims[i];
        const std::vector<int>& r = nk->radices[static_cast<size_t>(i - s0)];
        di.length = desc.lengths[i];
        di.tier = PFFT_TIER_WORKGROUP;
        di.n_factors = static_cast<int>(std::min<size_t>(r.size(), PFFT_MAX_FACTORS));
        for (int f = 0; f < di.n_factors; ++f) di.factors[f] = r[static_cast<size_t>(f)];
        di.workgroup_size = nk->k.wg;
        di.ffts_per_workgroup = nk->k.fpw;
        di.lds_bytes = nk->k.lds_bytes;
      }
      fused_from = s0;
    }
    if (fused_from == 0) return;
    long long inner = 1;
    // Two-pass plan for the last two dimensions (stockham_rows2d.hpp): pass 1 = whole rows + the first radix-RC
    // butterfly of the columns (contiguous rows on both sides), pass 2 = the remaining (n0 / RC)-point column FFTs
    // as a batch-interleaved transform over RC * n1 adjacent columns.  C5 (fp32 1024 x 1024 x 256): 1.57 -> 1.40 ms.
    if (fused_from == rank) {
      const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
      const long long n0 = static_cast<long long>(desc.lengths[rank - 2]);
      const long long n1 = static_cast<long long>(desc.lengths[rank - 1]);
      const long long nmat = B * (total / (n0 * n1));
      // chunk of matrices whose intermediate fits the Infinity Cache (cache_chunk_bytes): pass 1 as "writer", pass 2
      // as "reader"; a single matrix beyond the cache keeps the streamed kernels and one launch per pass
      const size_t matrix_bytes = static_cast<size_t>(n0) * static_cast<size_t>(n1) * elem_bytes();
      // (random data, tools/perf_cache.py: 1024^2 x 16 / 32 / 64 / 256 +6 / +13 / +7 / +2 %, 512^2 x 128 +9 %; a 32 MiB
      //  batch is 18 % faster with the streamed kernels, hence the lower bound)
      // (SPLIT_COMPLEX storage: the streamed kernels, one launch per pass)
      const bool cached = !split && cache_chunk_bytes() >= matrix_bytes &&
                          matrix_bytes * static_cast<size_t>(nmat) >= cache_chunk_bytes() / 2;
      const long long chunk_mats = cached ? even_chunks(std::max<long long>(1, std::min<long long>(
                                                            nmat, static_cast<long long>(cache_chunk_bytes() / matrix_bytes))),
                                                        nmat)
                                          : nmat;
      const rows2d_kernel* rk = find_rows2d(n1, n0, cached ? 1 : 0, split);
      const bool range_ok = static_cast<unsigned long long>(n0) * static_cast<unsigned long long>(n1) * elem_bytes() <
                            0xFFFFFFF0ull;
      const size_t all_bytes = static_cast<size_t>(B) * static_cast<size_t>(total) * elem_bytes();
      const bool alias_ok = desc.placement != PFFT_IN_PLACE || all_bytes <= global_chunk_bytes();
      if (rk != nullptr && range_ok && alias_ok) {
        const long long m = n0 / rk->rc;
        const long long cols = static_cast<long long>(rk->rc) * n1;
        addressing a{static_cast<long long>(vout.offset), cols, 1, n0 * n1};
        std::vector<stage> tail;
        pfft_dim_info_t di{};
        tail_policy = cached ? 2 : 0;
        // the attempt is planned into temporaries: when it is rejected below, whatever plan_1d reserved for it (scratch
        // of a two-stage column plan, chunk groups) is given back (its small twiddle tables stay uploaded
        // until the plan goes away)
        const size_t saved_scratch = scratch_bytes, saved_half = overlap_scratch_half;
        const int saved_groups = n_chunk_groups;
        const int tier = plan_1d(tail, m, nmat * cols, cols, BUF_OUT, a, BUF_OUT, a, false, scale, backward, &di);
        tail_policy = 0;
        const bool accepted = tier == PFFT_TIER_WORKGROUP && tail.size() == 1 && tail[0].strided != nullptr;
        if (!accepted) {
          scratch_bytes = saved_scratch;
          overlap_scratch_half = saved_half;
          n_chunk_groups = saved_groups;
        }
        if (accepted) {
          st.push_back(make_rows2d_stage(rk, nmat, n0, static_cast<long long>(vin.offset),
                                         static_cast<long long>(vout.offset), backward));
          tail[0].alias_scratch = 2;
          if (chunk_mats < nmat) {  // pass 1 and pass 2 advance together, chunk_mats matrices at a time
            const int group_id = n_chunk_groups++;
            stage& s1 = st.back();
            s1.chunk_group = group_id;
            s1.chunk_batches = chunk_mats;
            s1.ffts_per_batch = n0;
            s1.in_batch_dist = s1.out_batch_dist = n0 * n1;
            tail[0].chunk_group = group_id;
            tail[0].chunk_batches = chunk_mats;
            tail[0].ffts_per_batch = cols;
            tail[0].in_batch_dist = tail[0].out_batch_dist = n0 * n1;
            regrid_for_chunk(s1, chunk_mats * n0);
            regrid_for_chunk(tail[0], chunk_mats * cols);
          }
          st.push_back(tail[0]);
          if (desc.placement == PFFT_IN_PLACE) {
            alias_scratch_bytes = std::max(alias_scratch_bytes, static_cast<size_t>(chunk_mats) * matrix_bytes);
          }
          two_pass_chunk_bytes = std::max(two_pass_chunk_bytes, static_cast<size_t>(chunk_mats) * matrix_bytes);
          if (record) {
            pfft_dim_info_t& d1 = info.dims[rank - 1];
            d1.length = static_cast<uint64_t>(n1);
            d1.tier = PFFT_TIER_WORKGROUP;
            d1.n_factors = rk->n_radices;
            for (int i = 0; i < rk->n_radices; ++i) d1.factors[i] = rk->radices[i];
            d1.workgroup_size = rk->wg;
            d1.ffts_per_workgroup = rk->rc;
            d1.lds_bytes = rk->lds_bytes;
            pfft_dim_info_t& d0 = info.dims[rank - 2];
            d0 = di;  // the column dimension: radix RC (fused into pass 1), then the factors of n0 / RC
            d0.length = static_cast<uint64_t>(n0);
            const int nf = std::min<int>(di.n_factors, PFFT_MAX_FACTORS - 1);
            d0.factors[0] = rk->rc;
            for (int i = 0; i < nf; ++i) d0.factors[i + 1] = di.factors[i];
            d0.n_factors = nf + 1;
          }
          fused_from = rank - 2;  // `inner` is accumulated over lengths[fused_from..rank) below
        }
      }
    }
    if (fused_from == rank) {
      const long long last = static_cast<long long>(desc.lengths[rank - 1]);
      addressing ia{static_cast<long long>(vin.offset), 1, last, 0};
      addressing oa{static_cast<long long>(vout.offset), 1, last, 0};
      const long long count = B * (total / last);
      plan_1d(st, last, count, count, BUF_IN, ia, BUF_OUT, oa, true, scale, backward,
              record ? &info.dims[rank - 1] : nullptr);
      inner = last;
      fused_from = rank - 1;
    } else {
      for (int i = fused_from; i < rank; ++i) inner *= static_cast<long long>(desc.lengths[i]);
    }
    for (int i = fused_from - 1; i >= 0; --i) {
      const long long n = static_cast<long long>(desc.lengths[i]);
      const long long outer_count = B * (total / (inner * n));
      addressing a{static_cast<long long>(vout.offset), inner, 1, inner * n};
      plan_1d(st, n, outer_count * inner, inner, BUF_OUT, a, BUF_OUT, a, false, 1.0, backward,
              record ? &info.dims[i] : nullptr);
      inner *= n;
    }
  }

  /// `forced_n1`: the first factor of the four-step split (measured_split's candidates; 0: the planner's rules)
  plan_t(const pfft_desc_t& d, hipStream_t s, long long forced_n1_ = 0) : desc(d), stream(s), forced_n1(forced_n1_) {
    validate(desc);
    hip_check(hipGetDevice(&device), "hipGetDevice");
    hipDeviceProp_t prop;
    hip_check(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    n_cus = prop.multiProcessorCount;
    max_lds = prop.sharedMemPerBlock;
    info.rank = desc.rank;
    info.n_compute_units = n_cus;
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
  plan_t(const plan_t& o)
      : desc(o.desc), stream(o.stream), device(o.device), n_cus(o.n_cus), max_lds(o.max_lds), tables(o.tables),
        scratch_bytes(o.scratch_bytes), twiddle_bytes(o.twiddle_bytes), alias_scratch_bytes(o.alias_scratch_bytes),
        two_pass_chunk_bytes(o.two_pass_chunk_bytes),
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
  plan_t& operator=(const plan_t&) = delete;

  /// intermediate of the two-pass 2-D plan when the caller's buffers alias: allocated at commit for IN_PLACE
  /// descriptors, on first use when an OUT_OF_PLACE plan is executed with in == out
  /// (the lazy path is serialised, refuses to allocate inside a stream capture -- hipMalloc is not capturable: run the
  ///  aliasing execute once before capturing, or commit the descriptor IN_PLACE)
  void ensure_alias_scratch() {
    if (alias_scratch != nullptr) return;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (alias_scratch != nullptr) return;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) {
      fail(PFFT_INVALID_CONFIGURATION,
           "an OUT_OF_PLACE plan executed with aliasing buffers needs an intermediate that cannot be allocated during "
           "stream capture: execute it once before capturing, or commit the descriptor IN_PLACE");
    }
    if (alias_scratch_bytes == 0) alias_scratch_bytes = two_pass_chunk_bytes;
    hip_check(hipMalloc(&alias_scratch, alias_scratch_bytes), "hipMalloc(2-D intermediate)");
  }

  /// run stage `s` for the user transforms [b0, b0 + nb) (chunked stages) or entirely (nb < 0)
  void run_stage(const stage& s, const void* in_re, const void* in_im, void* out_re, void* out_im, long long b0 = 0,
                 long long nb = -1, const launch_ctx& lc = launch_ctx()) {
    const bool on_aux = lc.on_aux, any_order = lc.any_order;
    const size_t scratch_shift = lc.scratch_shift;
    const long long in_shift = nb < 0 ? 0 : b0 * s.in_batch_dist;    // elements
    const long long out_shift = nb < 0 ? 0 : b0 * s.out_batch_dist;  // elements
    const long long count = nb < 0 ? s.count : nb * s.ffts_per_batch;
    const size_t sb = static_cast<size_t>(scalar_bytes());
    const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
    hipStream_t stream = on_aux ? aux_stream : this->stream;
    char* const scratch = this->scratch == nullptr ? nullptr : static_cast<char*>(this->scratch) + scratch_shift;
    // resolve buffers: user buffers follow the descriptor's storage, scratch is always interleaved
    auto base_re = [&](int buf, bool is_in) -> const char* {
      if (buf == BUF_SCRATCH) return static_cast<const char*>(scratch);
      if (buf == BUF_IN) return static_cast<const char*>(in_re);
      (void)is_in;
      return static_cast<const char*>(out_re);
    };
    auto base_im = [&](int buf) -> const char* {
      if (buf == BUF_SCRATCH) return static_cast<const char*>(scratch) + sb;
      if (buf == BUF_IN) return split ? static_cast<const char*>(in_im) : static_cast<const char*>(in_re) + sb;
      return split ? static_cast<const char*>(out_im) : static_cast<const char*>(out_re) + sb;
    };
    auto step_of = [&](int buf) { return (buf == BUF_SCRATCH || !split) ? 2 : 1; };
    // (split storage: either pair of planes aliasing means the pass cannot write its output over its input)
    const bool aliased = s.alias_scratch != 0 && (in_re == out_re || (split && in_im != nullptr && in_im == out_im));
    if (aliased) ensure_alias_scratch();
    if (s.xcd != nullptr) {  // one launch for the whole batch: stage A and stage B tasks from per-XCD queues
      xcd_args x = s.xa;
      x.a.in = static_cast<const char*>(in_re) + static_cast<size_t>(s.in_addr.offset) * elem_bytes();
      x.a.out = scratch;
      x.b.in = scratch;
      x.b.out = static_cast<char*>(out_re) + static_cast<size_t>(s.out_addr.offset) * elem_bytes();
      x.ctl = static_cast<unsigned*>(xcd_ctl);
      x.tmap = static_cast<unsigned*>(xcd_tmap);
      x.report = xcd_report;
      // The persistent launch, then its recovery launch: every work-group of the latter reads one word and leaves unless a
      // hand-off wait of the former gave up, in which case it recomputes what is missing IN STREAM ORDER -- the
      // submission's event (the stop event of the last launch) and everything queued behind the execute see valid data.
      // Aliasing buffers: stage B from the slot rings first (the input of those transforms is already overwritten).
      const hipEvent_t stop = take_stop_event();
      hip_check(s.xcd->launch(stream, s.grid, s.lds_bytes, x, s.backward), "kernel launch");
      const bool aliasing = in_re == out_re;
      if (aliasing) {
        hip_check(s.xcd->launch_recover(stream, s.recover_grid, s.lds_bytes, x, s.backward, XCD_RECOVER_STAGE_B), "kernel launch");
      }
      if (stop != nullptr) arm_stop_event(stop);
      hip_check(s.xcd->launch_recover(stream, s.recover_grid, s.lds_bytes, x, s.backward,
                                      aliasing ? XCD_RECOVER_REST : XCD_RECOVER_ALL),
                "kernel launch");
      if (xcd_check) check_xcd_recoveries();
      return;
    }
    if (s.rows2d != nullptr) {
      rows2d_args a = s.ra;
      a.any_order = any_order ? 1 : 0;
      if (nb >= 0) a.nmat = nb;  // chunked: matrices [b0, b0 + nb)
      const size_t unit = split ? sb : elem_bytes();
      const size_t io = static_cast<size_t>(s.in_offset + in_shift) * unit;
      const size_t oo = static_cast<size_t>(s.out_offset + out_shift) * unit;
      a.in = static_cast<const char*>(in_re) + io;
      a.out = aliased ? static_cast<char*>(alias_scratch)  // one chunk at a time goes through the scratch
                      : static_cast<char*>(out_re) + oo;
      if (split) {  // the scratch of an in-place execute holds the two planes one after the other
        a.in_im = static_cast<const char*>(in_im) + io;
        a.out_im = aliased ? static_cast<char*>(alias_scratch) + alias_scratch_bytes / 2 : static_cast<char*>(out_im) + oo;
      }
      const long long groups = a.nmat * (a.n0 / s.rows2d->rc);
      unsigned grid = static_cast<unsigned>(std::min<long long>(s.grid, std::max<long long>(groups, 1)));
      hip_check(s.rows2d->launch != nullptr
                    ? (split ? s.rows2d->launch_split : s.rows2d->launch)(stream, grid, a, s.backward)
                    : jit_launch_rows2d(s.rows2d, stream, grid, a, s.backward),
                "kernel launch");
      return;
    }
    if (s.strided != nullptr) {
      strided_args a = s.sa;
      a.any_order = any_order ? 1 : 0;
      a.total = count;
      const long long groups = strided_groups(count, a.inner, s.strided->fpw);
      unsigned grid = static_cast<unsigned>(std::min<long long>(s.grid, std::max<long long>(groups, 1)));
      // column-shaped input in segments narrower than a 128-byte line (fp32 n = 2048 stages: 8 columns; the planes of
      // SPLIT_COMPLEX data at 16 fp32 / 8 fp64 columns): neighbouring groups on one XCD (strided_args::pair_xcd).
      // Only the plain strided kernel forms honour it; the grid becomes a multiple of 16.
      {
        const bool in_user_split = split && s.in_buf != BUF_SCRATCH;
        const size_t seg = static_cast<size_t>(s.strided->fpw) * (in_user_split ? sb : elem_bytes());
        const bool column_in = a.in_fdist == 1 && a.in_gdist == 0 && a.in_stride > 1;
        if (column_in && seg < 128 && s.row_mode == 0 && s.tiled_in == 0 && s.strided->fpw > 1 && grid >= 32 &&
            groups >= 32 && pair_xcd_enabled()) {
          a.pair_xcd = 1;
          grid &= ~15u;
        }
      }
      if (split && (s.in_buf == BUF_SCRATCH) != (s.out_buf == BUF_SCRATCH)) {  // mixed storage (four-step stages)
        const bool in_user = s.in_buf != BUF_SCRATCH;
        const size_t iu = in_user ? sb : elem_bytes(), ou = in_user ? elem_bytes() : sb;
        const size_t io = static_cast<size_t>(s.in_addr.offset + in_shift) * iu;
        const size_t oo = static_cast<size_t>(s.out_addr.offset + out_shift) * ou;
        a.in = base_re(s.in_buf, true) + io;
        a.in_im = in_user ? base_im(s.in_buf) + io : nullptr;
        a.out = const_cast<char*>(base_re(s.out_buf, false)) + oo;
        a.out_im = in_user ? nullptr : const_cast<char*>(base_im(s.out_buf)) + oo;
        if (s.tiled_in != 0 && !in_user) {
          hip_check(jit_launch_strided_mixed_tin(s.strided, stream, grid, a, s.backward), "kernel launch");
          return;
        }
        hip_check(s.row_mode == 1 && !in_user ? jit_launch_strided_row_mixed(s.strided, stream, grid, a, s.backward)
                                              : jit_launch_strided_mixed(s.strided, stream, grid, a, s.backward, in_user ? 2 : 3),
                  "kernel launch");
        return;
      }
      if (split && s.in_buf != BUF_SCRATCH) {  // both sides are user buffers
        const size_t io = static_cast<size_t>(s.in_addr.offset + in_shift) * sb;
        const size_t oo = static_cast<size_t>(s.out_addr.offset + out_shift) * sb;
        a.in = base_re(s.in_buf, true) + io;
        a.in_im = base_im(s.in_buf) + io;
        if (aliased && s.alias_scratch == 2) {  // two-pass 2-D plan, in-place execute: pass 1 left planes in the scratch
          a.in = alias_scratch;
          a.in_im = static_cast<const char*>(alias_scratch) + alias_scratch_bytes / 2;
        }
        a.out = const_cast<char*>(base_re(s.out_buf, false)) + oo;
        a.out_im = const_cast<char*>(base_im(s.out_buf)) + oo;
        hip_check(s.strided->launch != nullptr ? s.strided->launch_split(stream, grid, a, s.backward)
                                               : jit_launch_strided_split(s.strided, stream, grid, a, s.backward, s.store_modifier),
                  "kernel launch");
        return;
      }
      a.in = base_re(s.in_buf, true) + static_cast<size_t>(s.in_addr.offset + in_shift) * elem_bytes();
      if (aliased && s.alias_scratch == 2) a.in = alias_scratch;  // two-pass 2-D plan, in-place execute
      a.out = const_cast<char*>(base_re(s.out_buf, false)) +
              static_cast<size_t>(s.out_addr.offset + out_shift) * elem_bytes();
      if (s.row_mode != 0 && s.store_modifier == 0) {
        hip_check(s.strided->launch_row != nullptr
                      ? s.strided->launch_row(stream, grid, a, s.backward, s.row_mode - 1)
                      : jit_launch_strided_row(s.strided, stream, grid, a, s.backward, s.row_mode - 1),
                  "kernel launch");
        return;
      }
      if (s.tiled_in != 0) {
        hip_check(s.tiled_in == 2 ? s.strided->launch_tin_w(stream, grid, a, s.backward)
                                  : s.strided->launch_tin(stream, grid, a, s.backward),
                  "kernel launch");
        return;
      }
      hip_check(s.strided->launch != nullptr
                    ? s.strided->launch(stream, grid, a, s.backward, s.store_modifier)
                    : jit_launch_strided(s.strided, stream, grid, a, s.backward, s.store_modifier),
                "kernel launch");
      return;
    }
    if (!s.generic && s.unpacked != nullptr) {
      const bool user_split = split && s.in_buf != BUF_SCRATCH;
      const size_t unit = user_split ? sb : elem_bytes();
      const char* i_re = base_re(s.in_buf, true) + static_cast<size_t>(s.in_offset) * unit;
      const char* i_im = base_im(s.in_buf) + static_cast<size_t>(s.in_offset) * unit;
      char* o_re = const_cast<char*>(base_re(s.out_buf, false)) + static_cast<size_t>(s.out_offset) * unit;
      char* o_im = const_cast<char*>(base_im(s.out_buf)) + static_cast<size_t>(s.out_offset) * unit;
      hip_check(jit_launch_unpacked(s.unpacked, user_split, stream, s.grid, i_re, i_im, o_re, o_im, s.tw, s.count,
                                    s.scale, s.backward, static_cast<unsigned>(s.in_addr.stride),
                                    static_cast<unsigned>(s.in_addr.dist_inner),
                                    static_cast<unsigned>(s.out_addr.stride),
                                    static_cast<unsigned>(s.out_addr.dist_inner)),
                "kernel launch");
      return;
    }
    if (!s.generic) {
      if (split) {  // spec stages only touch user buffers when the storage is split (plan_1d)
        const size_t io = static_cast<size_t>(s.in_offset) * sb, oo = static_cast<size_t>(s.out_offset) * sb;
        const char* sr = static_cast<const char*>(s.in_buf == BUF_IN ? in_re : out_re);
        const char* si = static_cast<const char*>(s.in_buf == BUF_IN ? in_im : out_im);
        auto launch_split = s.spec->launch != nullptr ? s.spec->launch_split : nullptr;
        hip_check(launch_split != nullptr
                      ? launch_split(stream, s.grid, sr + io, si + io, static_cast<char*>(out_re) + oo,
                                     static_cast<char*>(out_im) + oo, s.tw, s.count, s.scale, s.backward)
                      : jit_launch_spec_split(s.spec, stream, s.grid, sr + io, si + io, static_cast<char*>(out_re) + oo,
                                              static_cast<char*>(out_im) + oo, s.tw, s.count, s.scale, s.backward),
                  "kernel launch");
        return;
      }
      const char* i = base_re(s.in_buf, true) + static_cast<size_t>(s.in_offset) * elem_bytes();
      char* o = const_cast<char*>(base_re(s.out_buf, false)) + static_cast<size_t>(s.out_offset) * elem_bytes();
      hip_check(s.spec->launch != nullptr
                    ? s.spec->launch(stream, s.grid, i, o, s.tw, s.count, s.scale, s.backward)
                    : jit_launch_spec(s.spec, stream, s.grid, i, o, s.tw, s.count, s.scale, s.backward),
                "kernel launch");
      return;
    }
    generic_args g = s.ga;
    g.total_count = count;
    g.in_step = step_of(s.in_buf);
    g.out_step = step_of(s.out_buf);
    const size_t ioff = static_cast<size_t>(s.in_addr.offset + in_shift);
    const size_t ooff = static_cast<size_t>(s.out_addr.offset + out_shift);
    g.in_re = base_re(s.in_buf, true) + ioff * sb * static_cast<size_t>(g.in_step);
    g.in_im = base_im(s.in_buf) + ioff * sb * static_cast<size_t>(g.in_step);
    g.out_re = const_cast<char*>(base_re(s.out_buf, false)) + ooff * sb * static_cast<size_t>(g.out_step);
    g.out_im = const_cast<char*>(base_im(s.out_buf)) + ooff * sb * static_cast<size_t>(g.out_step);
    const hipError_t e = desc.precision == PFFT_PRECISION_F64 ? launch_generic_f64(stream, s.grid, s.lds_bytes, g)
                                                              : launch_generic_f32(stream, s.grid, s.lds_bytes, g);
    hip_check(e, "kernel launch");
  }

  /// a two-launch chunk group with several chunks whose chunks do not share an intermediate buffer
  bool overlappable(const std::vector<stage>& st, size_t i, size_t j, bool several_chunks, bool aliased) const {
    if (!chunk_overlap_enabled() || j - i != 2 || !several_chunks) return false;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) {
      return false;  // a captured execute is a plain chain of kernel nodes
    }
    const stage& a = st[i];
    const stage& b = st[i + 1];
    if (aliased && (a.alias_scratch != 0 || b.alias_scratch != 0)) return false;  // one scratch chunk for all chunks
    const bool uses_scratch = a.in_buf == BUF_SCRATCH || a.out_buf == BUF_SCRATCH || b.in_buf == BUF_SCRATCH ||
                              b.out_buf == BUF_SCRATCH;
    return !uses_scratch || overlap_scratch_half > 0;
  }

  /// chunk c: first launch on the plan's stream, second launch on aux_stream behind an event; the plan's stream joins
  /// at the end.  The scratch (four-step plans) alternates between its two halves.
  void run_chunks_overlapped(const stage& a, const stage& b, long long batches, long long chunk_batches,
                             const void* in_re, const void* in_im, void* out_re, void* out_im) {
    if (overlap_mode == 2) {
      // Every second launch keeps its barrier, so when the barrier-free first launch of chunk c starts, only the
      // second launch of chunk c - 1 can still be running: other matrices, the other half of the scratch.
      size_t c = 0;
      for (long long b0 = 0; b0 < batches; b0 += chunk_batches, ++c) {
        const long long nb = std::min(chunk_batches, batches - b0);
        launch_ctx lc;
        lc.scratch_shift = (c & 1) * overlap_scratch_half;
        lc.any_order = c > 0;
        run_stage(a, in_re, in_im, out_re, out_im, b0, nb, lc);
        lc.any_order = false;
        run_stage(b, in_re, in_im, out_re, out_im, b0, nb, lc);
      }
      return;
    }
    if (aux_stream == nullptr) {
      hip_check(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking), "hipStreamCreate");
    }
    const size_t n_chunks = static_cast<size_t>((batches + chunk_batches - 1) / chunk_batches);
    while (chunk_events.size() < 2 * n_chunks) {  // [c]: first launch of chunk c done, [n_chunks + c]: second launch
      hipEvent_t ev = nullptr;
      hip_check(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
      chunk_events.push_back(ev);
    }
    size_t c = 0;
    for (long long b0 = 0; b0 < batches; b0 += chunk_batches, ++c) {
      const long long nb = std::min(chunk_batches, batches - b0);
      launch_ctx lc;
      lc.scratch_shift = (c & 1) * overlap_scratch_half;
      if (c >= 2 && overlap_scratch_half > 0) {  // this half of the scratch was last read by chunk c - 2
        hip_check(hipStreamWaitEvent(stream, chunk_events[n_chunks + c - 2], 0), "hipStreamWaitEvent");
      }
      run_stage(a, in_re, in_im, out_re, out_im, b0, nb, lc);
      hip_check(hipEventRecord(chunk_events[c], stream), "hipEventRecord");
      hip_check(hipStreamWaitEvent(aux_stream, chunk_events[c], 0), "hipStreamWaitEvent");
      lc.on_aux = true;
      run_stage(b, in_re, in_im, out_re, out_im, b0, nb, lc);
      if (c + 1 == n_chunks || overlap_scratch_half > 0) {
        hip_check(hipEventRecord(chunk_events[n_chunks + c], aux_stream), "hipEventRecord");
      }
    }
    hip_check(hipStreamWaitEvent(stream, chunk_events[2 * n_chunks - 1], 0), "hipStreamWaitEvent");
  }

  /// `completion`: the submission's completion event.  Returns true when it rode on the last launch as that dispatch's
  /// stop event (kernels.hpp: arm_stop_event); otherwise the caller records it behind the launches.
  bool execute(int direction, const void* in_re, const void* in_im, void* out_re, void* out_im,
               hipEvent_t completion = nullptr) {
    if (direction != PFFT_FORWARD && direction != PFFT_BACKWARD) {
      fail(PFFT_INVALID_CONFIGURATION, "Invalid direction ", direction);
    }
    if (in_re == nullptr || out_re == nullptr) fail(PFFT_INVALID_CONFIGURATION, "null data pointer");
    device_guard dg(device);  // launches go to the device the plan was committed on, whatever is current
    const std::vector<stage>& st = stages[direction];
    bool rode = false;
    if (completion != nullptr && !st.empty() && st.back().chunk_group < 0 && stop_event_on_launch()) {
      hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) {
        completion = nullptr;  // a captured execute records its event as a node of its own
      }
    } else {
      completion = nullptr;
    }
    for (size_t i = 0; i < st.size();) {
      if (st[i].chunk_group < 0) {
        if (completion != nullptr && i + 1 == st.size()) {
          arm_stop_event(completion);
          struct disarm {  // also when the launch throws
            bool* rode;
            ~disarm() { *rode = take_stop_event() == nullptr; }
          } guard{&rode};
          run_stage(st[i], in_re, in_im, out_re, out_im);
        } else {
          run_stage(st[i], in_re, in_im, out_re, out_im);
        }
        ++i;
        continue;
      }
      size_t j = i;
      while (j < st.size() && st[j].chunk_group == st[i].chunk_group) ++j;
      const long long batches = st[i].count / st[i].ffts_per_batch;
      const long long chunk_batches = std::max<long long>(1, st[i].chunk_batches);
      if (overlappable(st, i, j, batches > chunk_batches, in_re == out_re || (in_im != nullptr && in_im == out_im))) {
        run_chunks_overlapped(st[i], st[i + 1], batches, chunk_batches, in_re, in_im, out_re, out_im);
        i = j;
        continue;
      }
      for (long long b0 = 0; b0 < batches; b0 += chunk_batches) {
        const long long nb = std::min(chunk_batches, batches - b0);
        for (size_t k = i; k < j; ++k) run_stage(st[k], in_re, in_im, out_re, out_im, b0, nb);
      }
      i = j;
    }
    return rode;
  }

  /// PFFT_XCD_CHECK=1 (tests, fixed at commit): wait for every XCD-local execute and raise when it needed its recovery
  /// launch -- a hand-off wait gave up.  Without the knob such an execute is simply recomputed in stream order and counted
  /// (pfft_plan_info_t::xcd_recoveries); the tests want to know that it does not happen on a healthy device.
  static bool xcd_check_enabled() {
    const char* e = getenv("PFFT_XCD_CHECK");
    return e != nullptr && std::atoi(e) != 0;
  }
  void check_xcd_recoveries() {
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) return;
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    const unsigned n = __atomic_load_n(xcd_report, __ATOMIC_RELAXED);
    if (n != xcd_recoveries_seen) {
      xcd_recoveries_seen = n;
      fail(PFFT_INTERNAL_ERROR, "XCD-local four-step launch: ", xcd_report[1], " hand-off waits gave up (first: site ",
           xcd_report[2], ", local transform ", xcd_report[3], ", wanted ", xcd_report[4], ", saw ", xcd_report[5],
           "); the execute was recomputed by its recovery launch (PFFT_XCD_CHECK=1 reports this as an error)");
    }
  }

  /// PFFT_PAIR_XCD=0: never pair narrow-segment groups on one XCD (A/B)
  static bool pair_xcd_enabled() {
    static const bool on = [] {
      const char* e = getenv("PFFT_PAIR_XCD");
      return e == nullptr || std::atoi(e) != 0;
    }();
    return on;
  }

  /// PFFT_STOP_EVENT_ON_LAUNCH=0: always record completion events with hipEventRecord (A/B, tools/latency.py)
  static bool stop_event_on_launch() {
    static const bool on = [] {
      const char* e = getenv("PFFT_STOP_EVENT_ON_LAUNCH");
      return e == nullptr || std::atoi(e) != 0;
    }();
    return on;
  }
};

}  // namespace pfa

struct pfft_plan_t {
  std::unique_ptr<pfa::plan_t> impl;
};

namespace {
/// Completion events are recycled: creating and destroying a hipEvent_t per submission costs more than the launch
/// of a small transform.  pfft_event_destroy returns the event to this pool (per device), new events come from it.
struct event_pool {
  std::mutex m;
  std::vector<std::pair<int, hipEvent_t>> free_list;  // (device, event)
  hipEvent_t get(int device) {
    {
      std::lock_guard<std::mutex> lock(m);
      for (size_t i = free_list.size(); i-- > 0;) {
        if (free_list[i].first == device) {
          hipEvent_t ev = free_list[i].second;
          free_list[i] = free_list.back();
          free_list.pop_back();
          return ev;
        }
      }
    }
    hipEvent_t ev = nullptr;
    const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipEventCreate: ", hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(m);
    owner[ev] = device;
    return ev;
  }
  /// true when the event was taken back (events not created here are destroyed by the caller)
  bool put(hipEvent_t ev) {
    std::lock_guard<std::mutex> lock(m);
    const auto it = owner.find(ev);
    if (it == owner.end()) return false;
    if (free_list.size() >= 1024) {
      owner.erase(it);
      return false;
    }
    free_list.emplace_back(it->second, ev);
    return true;
  }
  std::map<hipEvent_t, int> owner;
};
event_pool& events() {
  static event_pool* p = new event_pool();  // never destroyed: events may outlive static destruction order
  return *p;
}

/// dependencies in, completion event out (shared by the two _ex entry points)
template <typename Run>
void execute_with_events(pfft_plan_t* plan, int32_t n_deps, void* const* deps, void** event_out, Run&& run) {
  pfa::plan_t& p = *plan->impl;
  pfa::device_guard dg(p.device);
  if (n_deps < 0 || (n_deps > 0 && deps == nullptr)) pfa::fail(PFFT_INVALID_CONFIGURATION, "invalid dependency list");
  for (int32_t i = 0; i < n_deps; ++i) {
    if (deps[i] == nullptr) continue;  // a default-constructed event: nothing to wait for
    const hipError_t e = hipStreamWaitEvent(p.stream, static_cast<hipEvent_t>(deps[i]), 0);
    if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipStreamWaitEvent: ", hipGetErrorString(e));
  }
  if (event_out == nullptr) {
    (void)run(nullptr);
    return;
  }
  *event_out = nullptr;
  hipEvent_t ev = events().get(p.device);
  bool rode = false;
  try {
    rode = run(ev);  // true: the event is the stop event of the last dispatch, no packet of its own
  } catch (...) {
    (void)events().put(ev);
    throw;
  }
  if (!rode) {
    const hipError_t e = hipEventRecord(ev, p.stream);
    if (e != hipSuccess) {
      (void)events().put(ev);
      pfa::fail(PFFT_HIP_ERROR, "hipEventRecord: ", hipGetErrorString(e));
    }
  }
  *event_out = ev;
}
}  // namespace

extern "C" {

pfft_status pfft_plan_create(const pfft_desc_t* desc, void* hip_stream, pfft_plan_t** plan) {
  return pfa::guarded([&] {
    if (desc == nullptr || plan == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null argument");
    *plan = nullptr;
    auto p = std::make_unique<pfft_plan_t>();
    p->impl = std::make_unique<pfa::plan_t>(*desc, static_cast<hipStream_t>(hip_stream));
    *plan = p.release();
  });
}

pfft_status pfft_plan_destroy(pfft_plan_t* plan) {
  return pfa::guarded([&] { delete plan; });
}

pfft_status pfft_plan_get_info(const pfft_plan_t* plan, pfft_plan_info_t* info) {
  return pfa::guarded([&] {
    if (plan == nullptr || info == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null argument");
    *info = plan->impl->info;
    if (plan->impl->xcd_report != nullptr) info->xcd_recoveries = __atomic_load_n(plan->impl->xcd_report, __ATOMIC_RELAXED);
  });
}

pfft_status pfft_execute(pfft_plan_t* plan, int32_t direction, const void* in, void* out) {
  return pfa::guarded([&] {
    if (plan == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null plan");
    if (plan->impl->desc.complex_storage != PFFT_INTERLEAVED_COMPLEX) {
      // committed_descriptor_impl.hpp:862-871
      pfa::fail(PFFT_INVALID_CONFIGURATION,
                "To use interleaved data layout, the descriptor.complex_storage must be INTERLEAVED_COMPLEX");
    }
    plan->impl->execute(direction, in, nullptr, out, nullptr);
  });
}

pfft_status pfft_execute_split(pfft_plan_t* plan, int32_t direction, const void* in_real, const void* in_imag,
                               void* out_real, void* out_imag) {
  return pfa::guarded([&] {
    if (plan == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null plan");
    if (plan->impl->desc.complex_storage != PFFT_SPLIT_COMPLEX) {
      pfa::fail(PFFT_INVALID_CONFIGURATION,
                "To use split data layout, the descriptor.complex_storage must be SPLIT_COMPLEX");
    }
    if (in_imag == nullptr || out_imag == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null imaginary pointer");
    plan->impl->execute(direction, in_real, in_imag, out_real, out_imag);
  });
}

pfft_status pfft_execute_ex(pfft_plan_t* plan, int32_t direction, const void* in, void* out, int32_t n_deps,
                            void* const* deps, void** event_out) {
  return pfa::guarded([&] {
    if (plan == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null plan");
    if (plan->impl->desc.complex_storage != PFFT_INTERLEAVED_COMPLEX) {
      pfa::fail(PFFT_INVALID_CONFIGURATION,
                "To use interleaved data layout, the descriptor.complex_storage must be INTERLEAVED_COMPLEX");
    }
    execute_with_events(plan, n_deps, deps, event_out,
                        [&](hipEvent_t ev) { return plan->impl->execute(direction, in, nullptr, out, nullptr, ev); });
  });
}

pfft_status pfft_execute_split_ex(pfft_plan_t* plan, int32_t direction, const void* in_real, const void* in_imag,
                                  void* out_real, void* out_imag, int32_t n_deps, void* const* deps,
                                  void** event_out) {
  return pfa::guarded([&] {
    if (plan == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null plan");
    if (plan->impl->desc.complex_storage != PFFT_SPLIT_COMPLEX) {
      pfa::fail(PFFT_INVALID_CONFIGURATION,
                "To use split data layout, the descriptor.complex_storage must be SPLIT_COMPLEX");
    }
    if (in_imag == nullptr || out_imag == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null imaginary pointer");
    execute_with_events(plan, n_deps, deps, event_out,
                        [&](hipEvent_t ev) {
                          return plan->impl->execute(direction, in_real, in_imag, out_real, out_imag, ev);
                        });
  });
}

pfft_status pfft_event_wait(void* event) {
  return pfa::guarded([&] {
    if (event == nullptr) return;
    const hipError_t e = hipEventSynchronize(static_cast<hipEvent_t>(event));
    if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipEventSynchronize: ", hipGetErrorString(e));
  });
}

pfft_status pfft_event_query(void* event, int32_t* done) {
  return pfa::guarded([&] {
    if (done == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null argument");
    *done = 1;
    if (event == nullptr) return;
    const hipError_t e = hipEventQuery(static_cast<hipEvent_t>(event));
    if (e == hipErrorNotReady) {
      *done = 0;
    } else if (e != hipSuccess) {
      pfa::fail(PFFT_HIP_ERROR, "hipEventQuery: ", hipGetErrorString(e));
    }
  });
}

pfft_status pfft_event_destroy(void* event) {
  return pfa::guarded([&] {
    if (event == nullptr) return;
    if (events().put(static_cast<hipEvent_t>(event))) return;  // recycled
    const hipError_t e = hipEventDestroy(static_cast<hipEvent_t>(event));
    if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipEventDestroy: ", hipGetErrorString(e));
  });
}

pfft_status pfft_queue_copy(void* hip_stream, const void* src, void* dst, size_t bytes, int32_t n_deps,
                            void* const* deps, void** event_out) {
  return pfa::guarded([&] {
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (n_deps < 0 || (n_deps > 0 && deps == nullptr)) pfa::fail(PFFT_INVALID_CONFIGURATION, "invalid dependency list");
    for (int32_t i = 0; i < n_deps; ++i) {
      if (deps[i] == nullptr) continue;
      const hipError_t e = hipStreamWaitEvent(st, static_cast<hipEvent_t>(deps[i]), 0);
      if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipStreamWaitEvent: ", hipGetErrorString(e));
    }
    if (bytes > 0) {
      const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st);
      if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipMemcpyAsync: ", hipGetErrorString(e));
    }
    if (event_out != nullptr) {
      *event_out = nullptr;
      int device = 0;
      (void)hipGetDevice(&device);
      hipEvent_t ev = events().get(device);
      const hipError_t e = hipEventRecord(ev, st);
      if (e != hipSuccess) {
        (void)events().put(ev);
        pfa::fail(PFFT_HIP_ERROR, "hipEventRecord: ", hipGetErrorString(e));
      }
      *event_out = ev;
    }
  });
}

pfft_status pfft_queue_wait(void* hip_stream) {
  return pfa::guarded([&] {
    const hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipStreamSynchronize: ", hipGetErrorString(e));
  });
}

pfft_status pfft_plan_clone(const pfft_plan_t* plan, pfft_plan_t** copy) {
  return pfa::guarded([&] {
    if (plan == nullptr || copy == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null argument");
    *copy = nullptr;
    auto p = std::make_unique<pfft_plan_t>();
    p->impl = std::make_unique<pfa::plan_t>(*plan->impl);
    *copy = p.release();
  });
}

pfft_status pfft_plan_wait(pfft_plan_t* plan) {
  return pfa::guarded([&] {
    if (plan == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null plan");
    const hipError_t e = hipStreamSynchronize(plan->impl->stream);
    if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipStreamSynchronize: ", hipGetErrorString(e));
  });
}

}  // extern "C"
