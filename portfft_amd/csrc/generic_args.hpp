// Launch-time arguments of the generic LDS tier (plain C++: shared by the host planner and the device kernel).
#pragma once

namespace pfa {

constexpr int GENERIC_MAX_PASSES = 12;
constexpr int GENERIC_WG = 256;

/// Launch-time description of one generic stage.  Offsets and strides are in complex elements; `step` is the
/// scalar step between consecutive complex elements of an array (2 interleaved, 1 split).
struct generic_args {
  const void* in_re;
  const void* in_im;
  void* out_re;
  void* out_im;
  int in_step, out_step;
  long long in_stride, out_stride;
  /// FFT number t lives at (t / inner_count) * dist_outer + (t % inner_count) * dist_inner
  long long in_dist_inner, in_dist_outer, out_dist_inner, out_dist_outer;
  long long inner_count, total_count;
  int n, n_passes;
  int radix[GENERIC_MAX_PASSES];
  int tw_off[GENERIC_MAX_PASSES];
  const void* tw;
  int fpw;                   // FFTs staged together by one work-group
  int in_f_fast, out_f_fast; // 1: consecutive lanes walk FFTs (batch-fastest), 0: consecutive lanes walk elements
  int conj_in, conj_out;
  double scale;
  /// optional store modifier: output element k of the FFT with inner index c is multiplied by W_M^{k*c},
  /// W_M^m = hi[m >> shift] * lo[m & ((1<<shift)-1)]  (two small tables instead of an M-entry one)
  const void* stw_lo;
  const void* stw_hi;
  int stw_shift;
  /// exact division by small runtime constants without integer-divide sequences:
  /// e / d == (e * magic(d)) >> 40 for e < 2^24, d < 2^16   (magic(d) = floor(2^40 / d) + 1)
  unsigned long long magic_n, magic_fpw;
  unsigned long long magic_nb[GENERIC_MAX_PASSES];
  unsigned long long magic_ns[GENERIC_MAX_PASSES];
};

inline unsigned long long generic_magic(unsigned d) { return ((1ull << 40) / d) + 1; }

/// radices the generic kernel can run; the planner factorises lengths into these (plan_core.cpp: choose_radices)
#define PFA_GENERIC_RADICES(X) \
  X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(19) X(23) X(29) X(31)
/// ... and the primes only the "big radix" instantiation of the kernel carries (generic_fft_kernel<T, true>): a
/// wave64 build of the reference takes any prime factor up to its sub-group size as one cross-lane DFT
/// (/root/reference/src/portfft/common/subgroup.hpp:226-253, CMakeLists.txt:54 PORTFFT_SUBGROUP_SIZES); here such a
/// factor is one in-register butterfly of the symmetric half-length form.  Kept out of the ordinary instantiation:
/// a radix-61 butterfly holds 61 complex values per lane, and the kernel's register count is that of its largest case.
#define PFA_GENERIC_RADICES_BIG(X) X(37) X(41) X(43) X(47) X(53) X(59) X(61)
constexpr int GENERIC_MAX_SMALL_RADIX = 31;


}  // namespace pfa
