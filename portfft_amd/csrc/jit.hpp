// Runtime specialisation of the work-group kernels (hiprtc).
//
// Role in the reference: portFFT specialises its kernels at commit time through SYCL specialization constants
// (/root/reference/src/portfft/committed_descriptor_impl.hpp:448-573, set_spec_constants) so that one source serves
// every length.  The MI355X analogue: lengths without a pre-compiled, hand-tuned instantiation (kernels_f32.hip, ...)
// get the same templates (stockham_wg.hpp / stockham_strided.hpp) instantiated at commit time by hiprtc for the
// radix sequence the planner below picks, so any length with prime factors <= 31 that fits LDS runs on a
// compile-time-radix kernel instead of the runtime-radix generic kernel (3-5x slower, profiles/r1_notes.md).
//
// The compiled code objects are cached per process and device, and on disk (PFFT_JIT_CACHE_DIR, default
// $XDG_CACHE_HOME/portfft_amd or ~/.cache/portfft_amd; empty string: no disk cache), keyed by a hash of the kernel
// sources, the instantiation, the architecture and the hiprtc version.
// PFFT_JIT=0 disables the facility (the planner then falls back to the generic tier); PFFT_JIT_VERBOSE=1 logs
// every compilation to stderr.
#pragma once
#include <string>
#include <vector>

#include "kernels.hpp"

namespace pfa {

/// the template arguments of wg_cfg (stockham_wg.hpp) chosen at run time
struct wg_params {
  int precision = 0;
  int n = 0;
  std::vector<int> radices;
  int wg = 0, fpw = 0, pads = 0, padw = 0, twm = 0, occ = 1, aux = 2, staged = 0, twl = 0;
  /// complex elements a lane holds in its widest pass
  int regs = 0;
};

/// Planner of the packed work-group tier for an arbitrary length: radix sequence (fewest LDS exchanges, balanced
/// radices), lanes per FFT (least idle lanes in ragged passes), FFTs per work-group, padding, I/O staging.
/// Returns false when `n` has a prime factor above 31 or does not fit `max_lds`.
/// forced_radices (measured planning, below): that radix sequence instead of the planner's own (product n, every radix a
/// supported butterfly, at most MAX_PASSES of them -- otherwise ignored)
bool choose_spec_params(int precision, long long n, size_t max_lds, wg_params* out,
                        const std::vector<int>* forced_radices = nullptr);

/// Measured planning (opt-in, PFFT_PLAN_MEASURE=1; the reference's rule is static, committed_descriptor_impl.hpp:210-313):
/// the radix sequences the planner ranks highest for a packed length, in the orders worth timing -- the planner's own
/// choice first.  plan.cpp times them on the plan's stream at commit and records the winner.
std::vector<std::vector<int>> spec_radix_candidates(int precision, long long n, size_t max_lds, int max_candidates = 14);
bool plan_measure_enabled();
/// the recorded choice for (arch, precision, n): process table, then the JIT cache directory
/// (`choice_<arch>_<f32|f64>_<n>.txt`, next to the code objects); empty when there is none
std::vector<int> plan_choice_lookup(const std::string& arch, int precision, long long n, int max_factor = 61);
/// (an empty sequence forgets the record)
void plan_choice_store(const std::string& arch, int precision, long long n, const std::vector<int>& radices);

/// Same for the strided tier (FPW adjacent FFTs side by side); `inner_count` is the number of adjacent FFTs the
/// stage offers (narrow stages get narrower groups).
/// want_fpw > 0: exactly that many FFTs per work-group or nothing (a four-step stage A that must match the group width
/// of its stage B, plan.cpp)
bool choose_strided_params(int precision, long long n, long long inner_count, size_t max_lds, wg_params* out,
                           bool column_both = false, int want_fpw = 0);

/// Planner of the first pass of the two-pass 2-D plan (stockham_rows2d.hpp) for rows of length n1 in a matrix of n0
/// rows: the column radix RC taken in that pass (= rows per work-group = wg_params::fpw; n0 % RC == 0), the row
/// radices (the last one is fused with the column radix into one 2-D butterfly: (n1 / r_last) % wg == 0) and the
/// lanes.  False when n1 has a prime factor above 31 or nothing fits the register / LDS budget.
/// rc_mask: the column radices (8 | 4 | 2) the caller can follow with a single column pass of length n0 / rc.
bool choose_rows2d_params(int precision, long long n1, long long n0, size_t max_lds, wg_params* out,
                          int rc_mask = 8 | 4 | 2);

/// Runtime-compiled stockham_rows2d_kernel for row length n1 (cached per device, precision, n1, column radix, cache
/// policy and storage); split: the SPLIT_COMPLEX form (rows2d_kernel::split)
const rows2d_kernel* jit_rows2d_kernel(int precision, long long n1, long long n0, size_t max_lds, std::string* why,
                                       int policy = 0, int split = 0, int rc_mask = 8 | 4 | 2);
hipError_t jit_launch_rows2d(const rows2d_kernel* k, hipStream_t stream, unsigned grid, const rows2d_args& args,
                             int backward);

/// LDS bytes of a packed work-group kernel with these parameters (wg_cfg::LDS_BYTES)
size_t spec_lds_bytes(const wg_params& p);

/// "pfa::wg_cfg<float, pfa::radix_list<..>, ...>"
std::string wg_cfg_type_name(const wg_params& p);

/// A fused N-D kernel (stockham_nd.hpp): `k` has the launch interface of a packed kernel of length prod(dims);
/// radices[d] are the passes of dimension d (their twiddle tables follow each other, last dimension first).
struct nd_kernel {
  spec_kernel k{};
  std::vector<int> dims;
  std::vector<std::vector<int>> radices;
  int pads = 0, padw = 0, occ = 1, regs = 0;
};

/// Planner of the fused N-D tier: all of the (rank >= 2) transform in LDS.  False when it does not fit (128 KiB)
/// or a dimension has a prime factor above 31.
bool choose_nd_params(int precision, const std::vector<long long>& dims, size_t max_lds, nd_kernel* out);

/// "pfa::nd_cfg<float, 4096, 256, 1, ..., pfa::nd_pass<...>, ...>"
std::string nd_cfg_type_name(const nd_kernel& p);

/// Runtime-compiled fused N-D kernel on the current device (cached), interleaved or split variant on demand.
const nd_kernel* jit_nd_kernel(int precision, const std::vector<long long>& dims, bool split, size_t max_lds,
                               std::string* why);

bool jit_enabled();

/// Runtime-compiled packed kernel for length n on the current device (cached); the requested storage variant
/// (interleaved or split) is compiled on demand.  nullptr + *why when the length cannot be planned or hiprtc fails.
/// plan_only: return the entry with its parameters without compiling the packed form (other forms of the same
/// configuration are built from it: jit_unpacked_kernel).
const spec_kernel* jit_spec_kernel(int precision, long long n, bool split, size_t max_lds, std::string* why,
                                   bool plan_only = false, const std::vector<int>* forced_radices = nullptr);
/// gfx name of the current device ("gfx950"), as the runtime compiler targets it
std::string jit_device_arch();

/// Runtime-compiled strided kernel; `store_modifier` / `split_mode` select the variant to make available
/// (split_mode: 0 interleaved, 1 split on both sides, 2 split input + store modifier (four-step stage A on
/// SPLIT_COMPLEX data), 3 split output (stage B); see stockham_strided.hpp).
const strided_kernel* jit_strided_kernel(int precision, long long n, long long inner_count, bool store_modifier,
                                         int split_mode, size_t max_lds, std::string* why,
                                         bool column_both = false, int policy = 0, int want_fpw = 0);

/// UNPACKED-layout form (stockham_wg_unpacked_kernel) of the packed configuration `like` (a pre-compiled or a
/// runtime-specialised entry): forward/backward module functions for interleaved or split storage.
struct unpacked_kernel {
  const spec_kernel* base = nullptr;
  hipFunction_t fn[2] = {nullptr, nullptr};        // interleaved
  hipFunction_t fn_split[2] = {nullptr, nullptr};  // split
};
const unpacked_kernel* jit_unpacked_kernel(const spec_kernel* like, bool split, std::string* why);
hipError_t jit_launch_unpacked(const unpacked_kernel* k, bool split, hipStream_t stream, unsigned grid, const void* in,
                               const void* in_im, void* out, void* out_im, const void* tw, long long nfft,
                               double scale, int backward, unsigned in_stride, unsigned in_dist, unsigned out_stride,
                               unsigned out_dist);

hipError_t jit_launch_spec(const spec_kernel* k, hipStream_t stream, unsigned grid, const void* in, void* out,
                           const void* tw, long long nfft, double scale, int backward);
hipError_t jit_launch_spec_split(const spec_kernel* k, hipStream_t stream, unsigned grid, const void* in_re,
                                 const void* in_im, void* out_re, void* out_im, const void* tw, long long nfft,
                                 double scale, int backward);
hipError_t jit_launch_strided(const strided_kernel* k, hipStream_t stream, unsigned grid, const strided_args& args,
                              int backward, int store_modifier);
hipError_t jit_launch_strided_split(const strided_kernel* k, hipStream_t stream, unsigned grid,
                                    const strided_args& args, int backward, int store_modifier = 0);
/// Make the row-staged form (stockham_strided_row_kernel; fp32, interleaved, no store modifier) of a
/// runtime-compiled strided entry available: row_out 0 = row-shaped input, 1 = row-shaped output.
/// split_mode 3 (row_out 0): the row-staged input form of the mixed stage B (interleaved scratch -> split planes),
/// launched with jit_launch_strided_row_mixed
bool jit_strided_ensure_row(const strided_kernel* k, int row_out, size_t max_lds, std::string* why, int split_mode = 0);
hipError_t jit_launch_strided_row_mixed(const strided_kernel* k, hipStream_t stream, unsigned grid,
                                        const strided_args& args, int backward);
/// Tiled-input form of the mixed stage B (interleaved group-major scratch -> split planes) of a runtime-compiled entry;
/// false when the entry's shape has no such form (stockham_strided.hpp: tin_supported)
bool jit_strided_ensure_mixed_tin(const strided_kernel* k, std::string* why);
hipError_t jit_launch_strided_mixed_tin(const strided_kernel* k, hipStream_t stream, unsigned grid,
                                        const strided_args& args, int backward);
hipError_t jit_launch_strided_row(const strided_kernel* k, hipStream_t stream, unsigned grid, const strided_args& args,
                                  int backward, int row_out);
hipError_t jit_launch_strided_mixed(const strided_kernel* k, hipStream_t stream, unsigned grid,
                                    const strided_args& args, int backward, int split_mode);

/// Compile (do not load) the forward + backward kernels of `p` for `arch`: needs no device, used by the build check
/// and the CPU tests.  kind 0: packed interleaved, 1: packed split, 2: strided, 3: strided with store modifier,
/// 4: first pass of the two-pass 2-D plan (p from choose_rows2d_params), 5: strided on split planes with the store
/// modifier (S1 of the three-stage plan), 6: mixed stage B (interleaved tiles -> planes) in its tiled-input form,
/// 7: interleaved tiled-input form (p must satisfy tin_supported: a half-pair stage B is registered, this is its template).
bool jit_compile_only(const wg_params& p, int kind, const char* arch, size_t* code_bytes, std::string* why);
bool jit_compile_only_nd(const nd_kernel& p, bool split, const char* arch, size_t* code_bytes, std::string* why);

/// counters for tests / plan info: kernels compiled by hiprtc and loaded from the disk cache in this process
void jit_stats(long long* compiled, long long* from_disk);

}  // namespace pfa
