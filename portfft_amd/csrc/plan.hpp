// Internal interface of the planner / executor behind pfft_plan_create / pfft_execute (the committed_descriptor of the
// reference).  Mirrors the role of /root/reference/src/portfft/committed_descriptor_impl.hpp:
//   ctor + prepare_implementation (:210-313, :716-768)      -> plan_t::plan_t / plan_1d                  plan_core.cpp, plan_global.cpp
//   calculate_twiddles (:430-434, per-level dispatchers)     -> host_twiddles (host long double -> device)
//   allocate_scratch_and_precompute_scan (:579-708)          -> scratch of the GLOBAL tier                plan_global.cpp
//   dispatch_direction / dispatch_dimensions (:852-950)      -> build_direction: a direction's stage list  plan_nd.cpp
//   run_kernel (:1088-1111)                                  -> run_stage / execute                        plan_exec.cpp
// The structure is our own: a plan is two flat lists of kernel launches ("stages"), one per direction, resolved at
// commit time; execute only binds the user pointers and enqueues them on the plan's stream.
// (The doc comment of a member function is at its definition; the declarations carry its first line.)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "descriptor.hpp"
#include "jit.hpp"
#include "kernels.hpp"

namespace pfa {

inline void hip_check(hipError_t e, const char* what) {
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
  int tiled_in = 0;  // 1: the kernel's tiled-input form (strided_kernel::launch_tin), 2: ... with tiles twice as wide
                     // (launch_tin_w)
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

constexpr double PI_L = 3.14159265358979323846264338327950288;

template <typename T>
inline std::vector<T> host_twiddles(const std::vector<int>& radices) {
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

/// offsets of the passes' tables inside host_twiddles' layout; the generic kernel's radix set; its factorisation of n
/// (plan_core.cpp)
std::vector<int> tw_offsets(const std::vector<int>& radices);
bool generic_radix_ok(int r);
std::vector<int> choose_radices(long long n);

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

/// Every environment knob of the planner and the executor, read ONCE when a plan is committed (a copy of a plan takes
/// its parent's): nothing in execute looks at the environment.  Almost all of them are A/B switches of a measured
/// decision -- the twin a parity test or a profile compares the default with -- or experiment overrides; the defaults are
/// the product.  `mask()` (pfft_plan_info_t::knob_mask) tells a test which of them a plan was committed under.
/// (The runtime compiler's own knobs -- PFFT_JIT*, PFFT_PLAN_MEASURE, PFFT_NO_TUNED_TABLE, ... -- are read by jit.cpp, also
/// at commit only.)
struct plan_knobs {
  // A/B switches (set = the alternative)
  bool no_precompiled = false, xlane = false, no_regres = false, no_ltw = false, no_stw_rowish = false;
  bool jit_spec_radices = false, no_mixed_rows = false, no_three_stage = false, debug_global_set = false;
  bool no_tiled_scratch = false, no_tiled_lanes = false, no_xcd_local = false, global_n1_set = false;
  bool tin_rows_gone = false;  // (PFFT_TIN_ROWS: round 5's opt-in row-lanes stage B, removed; the slot keeps the mask's bit order)
  bool nd_two_stage_columns = false, no_fs_pairs = false, no_half_pairs = false, no_split_rule = false;
  bool no_split_tiled = false, no_wide_tiles = false, two_pass_2d_off = false, jit_verbose = false;
  bool split_cached = true, pair_xcd = true, stop_event_on_launch = true, xcd_check = false;
  bool xcd_contig = true;  // PFFT_XCD_CONTIG=0: no XCD-contiguous group walk for stages with unaligned row pitches
  bool no_split_2d_cached = false;  // PFFT_NO_SPLIT_2D_CACHED=1: the two-pass 2-D plan of SPLIT_COMPLEX data streamed, one launch per pass (round 5)
  bool hx_over_registered = true;  // PFFT_HX_OVER_REGISTERED=0: a registered one-per-CU fp64 strided entry is not replaced by a register-resident plan
  // overrides (unset: -1 / 0 / empty)
  int chunk_overlap = 2, jit_groups_per_wg = -1, groups_per_wg = 0, xcd_slots = 0, xcd_lag = 0;
  int row_in_max_n = 512;  // PFFT_ROW_IN_MAX_N: longest stage whose row-shaped INPUT is staged through LDS
  bool groups_per_wg_set = false, global_chunk_mib_set = false, cache_chunk_mib_set = false;
  long global_chunk_mib = 0, cache_chunk_mib = 0;
  long long bi_n1 = 0;  // PFFT_BI_N1 (experiment): first factor of the two-stage batch-interleaved plan
  bool no_split_unaligned_policy = false;  // PFFT_NO_SPLIT_UNALIGNED_POLICY=1: policy 3 for interleaved data only (SPLIT_COMPLEX batch-interleaved planes stay streamed, as before the round's last session)
  bool no_unaligned_policy = false;  // PFFT_NO_UNALIGNED_POLICY=1: streamed accesses also for stages with unaligned row pitches (round 5)
  int bi_wide_fpw = 0;  // PFFT_BI_WIDE_FPW (experiment): group width of the one-pass plan of long batch-interleaved transforms
  bool no_bi_wide_split2 = false;  // PFFT_NO_BI_WIDE_SPLIT2=1: SPLIT_COMPLEX batch-interleaved N = 513 ... 1024 stay on 16 / 8 columns (the twin of the double-width wide groups)
  bool no_bi_wide = false;  // PFFT_NO_BI_WIDE=1: batch-interleaved lengths beyond the LDS at full group width stay on the two-stage plan (the twin of round 6's one-pass plan)
  bool no_big_bi = false;  // PFFT_NO_BIG_BI=1: batch-interleaved arrays of 4 GiB and more as in round 5 (narrow groups / generic tier)
  bool no_bi_n1_rule = false;  // PFFT_NO_BI_N1_RULE=1: the balanced split of rounds 2-5 for the two-stage batch-interleaved plan
  long long global_n1 = 0, three_stage_min = 0, three_stage_n3 = 0, xcd_min_batch = -1, xcd_max_iters = -1;
  std::string debug_global, global_layout;
  static plan_knobs from_env();
  /// bit i: the i-th knob of from_env()'s table differs from its default
  unsigned long long mask = 0;
};

struct plan_t {
  const plan_knobs kn = plan_knobs::from_env();  // (first member: the initialisers below read it)
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
  bool xcd_recovery_warned = false;  // the one-time stderr note about a recovered launch has been printed
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
  /// How consecutive chunks of a two-launch plan overlap (PFFT_CHUNK_OVERLAP, fixed at commit):
  /// 0 not at all: every launch in order on the plan's stream;
  /// 1 second launches on a second stream behind events (measured 20 % slower: profiles/r2_notes.md);
  /// 2 (default) the first launch of chunk c + 1 is enqueued without the in-order barrier (hipExtAnyOrderLaunch), so
  ///   it fills the tail of the second launch of chunk c (measured: C3 with cache-sized chunks +3.7 %, C5 / ref65536
  ///   +1.2 %).  A runtime that ignores the flag runs them in order.
  const int overlap_mode = kn.chunk_overlap;
  bool chunk_overlap_enabled() const { return overlap_mode != 0; }
  int scalar_bytes() const { return desc.precision == PFFT_PRECISION_F64 ? 8 : 4; }
  size_t elem_bytes() const { return 2 * static_cast<size_t>(scalar_bytes()); }
  ~plan_t();
  /// the control block of the XCD-local launch: all zero before its first launch (the kernel keeps it that way)
  void alloc_xcd_ctl();
  void* upload(const void* host, size_t bytes);
  void* upload_twiddles(const std::vector<int>& radices);
  /// fused N-D kernel: the per-dimension tables one after the other, last dimension first (nd_cfg_type_name)
  void* upload_nd_twiddles(const nd_kernel& nk);
  /// W_M^m split in two tables (see generic_args::stw_*)
  void upload_store_twiddles(long long M, int shift, const void** lo, const void** hi);
  /// levels / shift of the store-modifier tables of an M-point plan behind kernel k's LDS: the fewest levels (fewest ...
  void store_table_shape(const strided_kernel* k, long long M, int* levels, int* shift) const;
  /// can stage kernel k carry the tables behind its LDS?  (always, for the kernels the planners produce: their own ...
  bool store_tables_fit(const strided_kernel* k, long long M) const;
  /// Store-modifier tables of a strided stage: L tables of 2^shift entries, table l = W_M^(i << (l * shift)), so that ...
  void attach_store_tables(stage& s, long long M, bool on_loads = false);
  /// device copy of the multi-level tables W_M^(i << (l * shift)), l < levels, i < 2^shift (cached per (M, shift))
  const void* store_tables_for(long long M, int levels, int shift);
  void finish_store_tables(stage& s, const strided_kernel* k, size_t total, bool on_loads);
  /// Width of the intermediate's tiles -- i.e. the group width its stage A must have -- when `fb` is the four-step ...
  static int pair_tile(const strided_kernel* fb, long long n2, bool wide);
  const spec_kernel* find_spec(long long n, bool allow_hx = true) const;
  /// column_both: the stage is column-shaped on both sides -> the wide-group entry of the length, when there is one ...
  const strided_kernel* find_strided(long long n, bool column_both = false, bool row_side = false,
                                     long long inner_count = -1, int policy = 0, bool store_modifier = false,
                                     int fs_stage = 0, bool allow_ltw = true) const;
  /// FFTs per work-group of the strided kernel get_strided(n, inner_count, ...) would deliver; 0 when there is none. ...
  int strided_fpw(long long n, long long inner_count) const;
  /// the pre-compiled strided kernel when it suits the stage, otherwise a runtime-specialised one (jit.hpp) ...
  const strided_kernel* get_strided(long long n, long long inner_count, bool store_modifier, bool user_split,
                                    bool column_both = false, bool row_side = false, int policy = 0);
  /// four-step stages on SPLIT_COMPLEX data: split user side, interleaved scratch side (runtime-specialised only)
  const strided_kernel* get_strided_mixed(long long n, long long inner_count, int split_mode, int policy = 0);
  /// PFFT_JIT_VERBOSE: say why a configuration stayed on the slower tier
  void jit_note(const char* what, long long n, const std::string& why) const;
  /// the pre-compiled packed kernel, otherwise a runtime-specialised one
  const spec_kernel* get_spec(long long n);
  /// Measured planning of the four-step split (PFFT_PLAN_MEASURE=1): every n1 x n2 with both factors in 32 ... 4096, no ...
  long long measured_split(long long n, long long count, long long static_n1);
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
  void fill_uniform(void* dst, size_t bytes);
  /// Measured planning (PFFT_PLAN_MEASURE=1; the reference's rule is static, committed_descriptor_impl.hpp:210-313): the ...
  std::vector<int> measured_radices(long long n);
  /// work-group loop trips of a strided stage (stockham_strided.hpp: strided_ngroups)
  static long long strided_groups(long long count, long long inner, int fpw);
  /// can the strided kernel `k` address this stage?  (interleaved data, whole groups, 32-bit byte ranges)
  bool strided_fits(const strided_kernel* k, long long inner_count, int in_buf, const addressing& ia, int out_buf,
                    const addressing& oa) const;
  const rows2d_kernel* find_rows2d(long long n1, long long n0, int policy, bool split = false);
  const rows2d_kernel* find_rows2d_registered(long long n1, long long n0, int policy, bool split) const;
  /// W_n^m for m in [0, n): the inter-pass column twiddles of the two-pass 2-D plan
  const void* upload_unit_roots(long long n);
  stage make_rows2d_stage(const rows2d_kernel* k, long long nmat, long long n0, long long in_off, long long out_off,
                          int backward);
  stage make_strided_stage(const strided_kernel* k, long long count, long long inner_count, int in_buf,
                           const addressing& ia, int out_buf, const addressing& oa, double scale, int backward,
                           int store_modifier = 0, bool allow_row = true);
  /// the launch grid of a chunked stage is sized for ONE chunk (`count` FFTs / `nmat` matrices), not for the whole ...
  void regrid_for_chunk(stage& s, long long count);
  /// `count` transforms in chunks of at most `chunk`: the same number of chunks, equally filled -- and one chunk fewer when ...
  static long long even_chunks(long long chunk, long long count);
  /// bytes of intermediate data per chunk of the GLOBAL tier = cap of the scratch allocation ...
  size_t global_chunk_bytes() const;
  /// Two-launch plans (four-step tier, two-pass 2-D plan) run chunk by chunk with the intermediate of a chunk sized ...
  size_t cache_chunk_bytes() const;
  /// largest length the generic tier can hold (two LDS images)
  long long generic_max_n() const { return static_cast<long long>(max_lds / (2 * elem_bytes())); }
  /// Grid of a persistent kernel.  Measured on the N=4096 kernel (tools/probes/proto_c2.hip, interleaved rounds): a grid of ...
  unsigned persistent_grid(const void* fn, hipFunction_t mfn, int wg, size_t lds, long long groups, int groups_per_wg);
  stage make_spec_stage(const spec_kernel* k, long long count, int in_buf, long long in_off, int out_buf,
                        long long out_off, double scale, int backward, const void* twiddles = nullptr,
                        const unpacked_kernel* unpacked = nullptr);
  stage make_generic_stage(long long n, long long count, long long inner_count, int in_buf, const addressing& ia,
                           int out_buf, const addressing& oa, double scale, int conj_in, int conj_out);
  /// BATCH_INTERLEAVED on both sides (element i of transform b at i * B + b), length n = n1 * n2, B transforms -- ...
  bool plan_batch_interleaved_two_stage(std::vector<stage>& out, long long n, long long B, long long outer, int in_buf,
                                        int out_buf, const addressing& ia, const addressing& oa, double scale,
                                        int backward, pfft_dim_info_t* info);
  /// Plan `count` 1-D FFTs of length n.  Returns the tier used. ...
  bool plan_three_stage(std::vector<stage>& out, long long n, long long count, const addressing& ia,
                        const addressing& oa, double scale, int backward, pfft_dim_info_t* info);
  /// XCC ids of the plan's device (census kernel, once per device and process); 0 when the census failed
  int xcd_queue_count();
  /// GLOBAL tier, XCD-local form (stockham_xcd.hpp; the reference keeps its batches-in-flight inside the last-level cache, ...
  bool plan_xcd_local(std::vector<stage>& out, long long n, long long count, const addressing& ia,
                      const addressing& oa, double scale, int backward, pfft_dim_info_t* info);
  int plan_1d(std::vector<stage>& out, long long n, long long count, long long inner_count, int in_buf,
              const addressing& ia, int out_buf, const addressing& oa, bool packed_io, double scale, int backward,
              pfft_dim_info_t* info);
  void build_direction(int direction);
  /// `forced_n1`: the first factor of the four-step split (measured_split's candidates; 0: the planner's rules)
  plan_t(const pfft_desc_t& d, hipStream_t s, long long forced_n1_ = 0);
  /// Copy of a committed plan (committed_descriptor_impl.hpp:774-817): the kernels and the twiddle tables are shared, ...
  plan_t(const plan_t& o);
  plan_t& operator=(const plan_t&) = delete;
  /// intermediate of the two-pass 2-D plan when the caller's buffers alias: allocated at commit for IN_PLACE ...
  void ensure_alias_scratch();
  /// run stage `s` for the user transforms [b0, b0 + nb) (chunked stages) or entirely (nb < 0)
  void run_stage(const stage& s, const void* in_re, const void* in_im, void* out_re, void* out_im, long long b0 = 0,
                 long long nb = -1, const launch_ctx& lc = launch_ctx());
  /// a two-launch chunk group with several chunks whose chunks do not share an intermediate buffer
  bool overlappable(const std::vector<stage>& st, size_t i, size_t j, bool several_chunks, bool aliased) const;
  /// chunk c: first launch on the plan's stream, second launch on aux_stream behind an event; the plan's stream joins ...
  void run_chunks_overlapped(const stage& a, const stage& b, long long batches, long long chunk_batches,
                             const void* in_re, const void* in_im, void* out_re, void* out_im);
  /// `completion`: the submission's completion event.  Returns true when it rode on the last launch as that dispatch's ...
  bool execute(int direction, const void* in_re, const void* in_im, void* out_re, void* out_im,
               hipEvent_t completion = nullptr);
  /// PFFT_XCD_CHECK=1 (tests, fixed at commit): wait for every XCD-local execute and raise when it needed its recovery ...
  void check_xcd_recoveries();
};

}  // namespace pfa
