// Registry of the pre-compiled gfx950 kernels (host-visible interface of kernels_f32.hip / kernels_f64.hip).
//
// The reference JIT-specialises its kernels at commit time through SYCL specialization constants
// (/root/reference/src/portfft/committed_descriptor_impl.hpp:448-573).  Here the hand-tuned variants are
// offline-compiled template instantiations that commit only looks up; every other length gets the same templates
// instantiated at commit time by hiprtc (jit.hpp) -- those entries carry module functions (mfn) instead of host
// symbols and launch pointers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <cstddef>

#include "generic_args.hpp"
#include "strided_args.hpp"
#include "xcd_args.hpp"

namespace pfa {

/// One specialised work-group kernel: packed, interleaved FFTs of a fixed length.
struct spec_kernel {
  int precision;  // PFFT_PRECISION_*
  int n;
  int wg;   // threads per work-group
  int fpw;  // FFTs per work-group
  size_t lds_bytes;
  int n_radices;
  int radices[8];
  int tw_total;  // complex entries of the twiddle table the kernel expects (layout: radix_list::tw_off)
  int tw_in_regs;  // 1: the kernel keeps its twiddles in VGPRs for its whole lifetime (TW_REGS)
  int groups_per_wg;  // tuned grid rule: FFT groups each work-group handles; 0 = persistent grid of 2x resident
  const void* fn[2];  // kernel symbols, [0] forward, [1] backward (for occupancy queries / attributes)
  hipError_t (*launch)(hipStream_t stream, unsigned grid, const void* in, void* out, const void* tw, long long nfft,
                       double scale, int backward);
  /// SPLIT_COMPLEX form (separate real / imaginary planes); fn_split are its kernel symbols
  const void* fn_split[2];
  hipError_t (*launch_split)(hipStream_t stream, unsigned grid, const void* in_re, const void* in_im, void* out_re,
                             void* out_im, const void* tw, long long nfft, double scale, int backward);
  /// runtime-compiled entries (jit.cpp): module functions, launched with jit_launch_spec*; fn / launch are null
  hipFunction_t mfn[2];
  hipFunction_t mfn_split[2];
  /// the remaining wg_cfg arguments, so that other forms of the same configuration (UNPACKED layouts) can be
  /// instantiated at run time
  int pads, padw, twm, occ, aux, staged, twl;
  /// 1: cross-lane variant of the length (stockham_xlane.hpp); only chosen when PFFT_XLANE is set (measurement:
  /// profiles/r2_notes.md)
  int xlane;
  /// 1: register-resident form (stockham_wg_hx.hpp): the transform does not fit LDS, lds_bytes is its half image; the
  /// wg_cfg fields above do not describe a configuration the other packed forms (UNPACKED layouts) could be built from
  int hx;
};

/// One strided work-group kernel (stockham_strided.hpp): FPW FFTs side by side, any element stride / FFT distance.
struct strided_kernel {
  int precision;
  int n;
  int wg;
  int fpw;
  size_t lds_bytes;
  int n_radices;
  int radices[8];
  int groups_per_wg;  // tuned grid rule (see spec_kernel)
  const void* fn[4];  // [backward * 2 + store_modifier]
  hipError_t (*launch)(hipStream_t stream, unsigned grid, const strided_args& args, int backward, int store_modifier);
  /// row-staged forms (stockham_strided_row_kernel): fn_row[row_out * 2 + backward]; null when not instantiated
  const void* fn_row[4];
  size_t lds_bytes_row;
  hipError_t (*launch_row)(hipStream_t stream, unsigned grid, const strided_args& args, int backward, int row_out);
  /// SPLIT_COMPLEX form on both sides (no store modifier); fn_split[backward]
  const void* fn_split[2];
  hipError_t (*launch_split)(hipStream_t stream, unsigned grid, const strided_args& args, int backward);
  /// runtime-compiled entries (jit.cpp): mfn[backward * 2 + store_modifier], mfn_split[backward]
  hipFunction_t mfn[4];
  hipFunction_t mfn_split[2];
  /// ... with the store modifier on SPLIT_COMPLEX user planes on both sides (S1 of the three-stage plan): [backward]
  hipFunction_t mfn_split_stw[2];
  /// mixed storage for the four-step tier on SPLIT_COMPLEX data: [backward] split input -> interleaved scratch with
  /// store modifier (stage A), [2 + backward] interleaved scratch -> split output (stage B)
  hipFunction_t mfn_mixed[4];
  /// row-staged forms of runtime-compiled entries: mfn_row[row_out * 2 + backward] (lds_bytes_row as above)
  hipFunction_t mfn_row[4];
  /// ... and the row-staged input form of the mixed stage B (interleaved scratch rows -> split planes): [backward]
  hipFunction_t mfn_row_mixed[2];
  /// ... and its tiled-input form (stockham_strided_kernel<Cfg, BWD, 0, 3, TIN = true>, jit_strided_ensure_mixed_tin):
  /// stage B behind a group-major intermediate on SPLIT_COMPLEX data
  hipFunction_t mfn_mixed_tin[2];
  /// tiled-input form (strided_pass TIN): the four-step stage B behind a group-major stage A of the same group
  /// width; fn_tin[backward]; null when not instantiated
  const void* fn_tin[2];
  hipError_t (*launch_tin)(hipStream_t stream, unsigned grid, const strided_args& args, int backward);
  /// ... and the same for tiles of tin_w = 2 * fpw elements (a stage A with groups twice as wide: fp32 n = 2048 holds 8
  /// columns, its stage A 16 -- 128-byte segments on stage A's side); null / 0 when not instantiated
  const void* fn_tin_w[2];
  hipError_t (*launch_tin_w)(hipStream_t stream, unsigned grid, const strided_args& args, int backward);
  int tin_w;
  /// cache policy of the entry's HBM accesses (stockham_wg.hpp, aux_of_loads / aux_of_stores): 0 everything streamed
  /// (nt), 1 "writer" (streamed loads, default-policy stores: fills an intermediate that should stay in the
  /// Infinity Cache), 2 "reader" (default-policy loads, streamed stores).  Policy twins carry the interleaved forms only.
  int policy;
  /// 1: alternative entry for the same length, preferred when both sides of the stage are column-shaped
  int wide;
  /// 1: alternative entry preferred when one side of the stage is row-shaped (its `_row` forms pay at this length)
  int rowish;
  /// where the store-modifier forms take their tables from (stockham_strided.hpp, STW): 1 small multi-level tables in
  /// LDS (every pre-compiled entry), 2 two global tables (runtime-specialised entries without LDS headroom)
  int stw_mode;
  /// four-step (GLOBAL tier) stage entries: fs_a = the entry of its length for stage A (store modifier, writes the
  /// group-major intermediate), fs_b = for stage B (tiled-input form, launch_tin).  A pair (fs_a, fs_b) with equal
  /// group widths replaces the default entries of the two lengths (tools/tune_fourstep.hip, profiles/r3_notes.md:
  /// narrow groups at two to four work-groups per CU beat wide ones at one).  fs_groups_per_wg: grid rule inside
  /// such a pair (0: groups_per_wg).
  int fs_a, fs_b, fs_groups_per_wg;
  int fs_only;  // 1: the entry exists only for such pairs (never the default entry of its length)
  /// 1 (fs_b entries): as stage B of a pair this entry carries the inter-stage twiddles on its LOADS (its tiled-input
  /// form multiplies the inputs of pass 0: strided_pass0_compute LTW) and stage A runs without the store modifier --
  /// fp64 n = 1024 (C3): stage A 120 -> 108 us per chunk, stage B 90 -> 91-94 (tools/tune_fourstep.hip case 120)
  int fs_ltw;
  /// > 0 (runtime-compiled entries): the register-resident form (stockham_strided_hx.hpp) planned as that many work-groups
  /// per CU; lds_bytes is its HALF image (+ TWL copy), the interleaved / split / mixed forms are that kernel's, the
  /// row-staged forms stay those of stockham_strided.hpp and there are no tiled-input forms
  int hx;
  /// 1 (runtime-compiled entries): the BIG forms (strided_io_big: groups that span 4 GiB or more); such an entry serves nothing else
  int big;
};

/// First pass of the two-pass 2-D plan (stockham_rows2d.hpp): whole row FFTs of length n + the first radix-rc
/// butterfly of the column FFT, rows {M*a + b} -> rows {rc*b + u}.
struct rows2d_kernel {
  int precision;
  int n;   // row length
  int rc;  // column radix taken in this pass (rows per work-group)
  int wg;
  size_t lds_bytes;
  int n_radices;
  int radices[8];
  int groups_per_wg;
  const void* fn[2];  // [backward]
  hipError_t (*launch)(hipStream_t stream, unsigned grid, const rows2d_args& args, int backward);
  /// runtime-compiled entries (jit.cpp): module functions [backward]; fn / launch are null
  hipFunction_t mfn[2];
  int policy;  // see strided_kernel::policy (0 or 1)
  /// SPLIT_COMPLEX storage on both sides (rows2d_args::in_im / out_im); null on the cache-policy twins
  const void* fn_split[2];
  hipError_t (*launch_split)(hipStream_t stream, unsigned grid, const rows2d_args& args, int backward);
  int split;  // runtime-compiled entries: mfn are the split-storage forms
};
const rows2d_kernel* rows2d_kernels(int* count);

/// XCD-local four-step kernel (stockham_xcd.hpp): both stages of an n1 x n2 transform in one persistent launch
struct xcd_kernel {
  int precision;
  int n1, n2;
  int wg, fpw;        // lanes of the launch's work-groups; columns (stage A) / rows (stage B) of a group
  int tasks_a, tasks_b;  // tickets of a transform per stage (a task = wg / stage-wg groups side by side)
  /// LDS layout (bytes from the dynamic base, xcd_layout): the leading twiddle tables of the two stages, the
  /// store-modifier tables; the launch adds those tables (16-byte rounded) and XCD_LDS_CTL_BYTES of control words
  unsigned twl_a_off, twl_b_off, stw_off;
  int n_radices_a, n_radices_b;
  int radices_a[8], radices_b[8];
  const void* fn[2];  // [backward]
  hipError_t (*launch)(hipStream_t stream, unsigned grid, size_t lds, const xcd_args& args, int backward);
  /// the recovery launch that follows every launch (stockham_xcd_recover_kernel; phase: XCD_RECOVER_*, xcd_args.hpp)
  const void* fn_recover[2];
  hipError_t (*launch_recover)(hipStream_t stream, unsigned grid, size_t lds, const xcd_args& args, int backward, int phase);
  int slots, lag, lookahead;  // tuned schedule (xcd_args)
  int wg_per_cu;              // work-groups per CU the launch is padded to (0: as many as fit)
  int min_mib;                // MiB of data per execute from which the launch beats the two-launch plan
};
const xcd_kernel* xcd_kernels(int* count);
/// XCC ids the device hands to work-groups (0: the census failed)
int xcd_census(hipStream_t stream);

/// AUX template values of the three cache policies (stockham_wg.hpp: loads in bits 0-7, stores + 1 in bits 8-15;
/// hardware bits 1 = sc0, 2 = nt, 16 = sc1).  The writer's stores carry sc1: written through the XCD's L2 instead of
/// left dirty in it until the end of the launch, still allocated in the Infinity Cache (tools/tune_2d_small.hip,
/// TUNE_STORE_POLICY: C5 in 256 MiB chunks 1405 us with plain stores, 1378 with sc1 or sc0|sc1, 1590 with nt).
enum : int { PFA_AUX_NT = 2, PFA_AUX_WRITER = ((16 + 1) << 8) | 2, PFA_AUX_READER = 0x300 };
/// (runtime-specialised kernels, jit.cpp; PFFT_JIT_WRITER_AUX overrides the writer's value for experiments)
inline int aux_of_policy(int policy) {
  if (policy == 1) {
    if (const char* e = getenv("PFFT_JIT_WRITER_AUX")) return static_cast<int>(std::strtol(e, nullptr, 0));
    return PFA_AUX_WRITER;
  }
  if (policy == 0) {
    if (const char* e = getenv("PFFT_JIT_NT_AUX")) return static_cast<int>(std::strtol(e, nullptr, 0));  // (experiments)
  }
  // policy 3 (round 6, kernels compiled at commit only): default policy on loads AND stores -- for stages whose column-shaped
  // side has a row pitch that is no multiple of a 128-byte line: every segment shares its first and last line with the
  // neighbouring group, and only lines that live in the L2 are fetched once and written back whole (with the XCD-contiguous
  // walk the neighbours run on one L2).  Streamed (nt) accesses left batch-interleaved N = 768 at a batch of 174 769 at 0.225
  // of the HBM peak against 0.616 at 174 768; with this policy 0.506 (profiles/r6_bi_unaligned_policies.txt).
  if (policy == 3) return 0;
  // (NOT for the four-step stages of lengths like 68640 = 104 x 660 or 10^6, whose pitches are unaligned too: on default policies
  //  -- both sides, or the user's side only -- they lose 10-26 %, profiles/r6_unaligned_policy.txt / r6_fs_unaligned.txt: there the
  //  streamed user side is what keeps the intermediate in the Infinity Cache)
  return policy == 2 ? PFA_AUX_READER : PFA_AUX_NT;
}

/// Completion event of the submission being enqueued (pfft_execute*_ex with event_out): plan_t::execute arms it in
/// front of its LAST launch, the launch helpers (kernels_impl.hpp, jit.cpp) take it and hand it to
/// hipExtLaunchKernel / hipExtModuleLaunchKernel as the dispatch's stop event -- no separate hipEventRecord packet
/// behind a small transform (tools/latency.py).  Thread-local; whoever armed it records the event the ordinary way
/// when no launch helper took it.
void arm_stop_event(hipEvent_t ev);
hipEvent_t take_stop_event();  // the armed event (and disarms), or nullptr

const strided_kernel* strided_kernels_f32(int* count);
const strided_kernel* strided_kernels_f64(int* count);

const spec_kernel* spec_kernels_f32(int* count);
const spec_kernel* spec_kernels_f64(int* count);

hipError_t launch_generic_f32(hipStream_t stream, unsigned grid, size_t lds_bytes, const generic_args& args);
hipError_t launch_generic_f64(hipStream_t stream, unsigned grid, size_t lds_bytes, const generic_args& args);
const void* generic_kernel_symbol(int precision, bool big_radix = false);

}  // namespace pfa
