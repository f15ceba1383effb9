// Launch-time arguments of the strided work-group tier (plain C++: shared by the host planner and the kernel).
#pragma once

namespace pfa {

/// Launch-time description of a strided stage (all counts in complex elements).
/// FFT t: (o, c) = (t / inner, t % inner); its element i lives at  o * dist_outer + c * fdist + i * stride.
struct strided_args {
  const void* in;  // interleaved complex, or the real plane when the kernel is a split-storage variant
  void* out;
  const void* in_im;  // imaginary planes (split-storage variants only)
  void* out_im;
  const void* tw;
  long long total;  // number of FFTs
  long long inner;  // FFTs per outer index (a group of FPW adjacent FFTs never straddles an outer index)
  long long in_dist_outer, out_dist_outer;
  unsigned in_stride, out_stride;  // element stride inside one FFT
  unsigned in_fdist, out_fdist;    // distance between consecutive FFTs of a group
  double scale;
  /// Store modifier W_M^(k * c) (STW kernels): stw_levels tables of 2^stw_lshift entries, contiguous at stw_tab;
  /// table l holds W_M^(i << (l * stw_lshift)), so W_M^m is the product of one entry per table.  The kernel copies
  /// them behind its own LDS once per work-group (the launch adds (stw_levels << stw_lshift) complex elements).
  const void* stw_tab;
  int stw_levels;
  int stw_lshift;
  long long stw_cdiv;  // store-modifier column index c = (inner index) / stw_cdiv  (1 for packed data)
  /// ... or (kernels instantiated with STW == 2) two global tables: W_M^m = stw_lo[m & (2^stw_shift - 1)] *
  /// stw_hi[m >> stw_shift]
  const void* stw_lo;
  const void* stw_hi;
  int stw_shift;
  /// group-major ("tiled") sides: when non-zero, group g of that side starts at g * gdist and (o, c) play no role in
  /// its address -- an intermediate written by one stage for the next can then be contiguous per work-group
  long long in_gdist, out_gdist;
  /// two-level element stride ("tiles" of 2^shift consecutive elements): element i of an FFT lives at
  /// (i >> shift) * stride + (i & (2^shift - 1)); shift 0 = plain element stride.  Requires the first-pass (input)
  /// / last-pass (output) butterfly stride of the kernel to be a multiple of the tile.
  int in_tile_shift, out_tile_shift;
  /// stride of the index inside an output tile (0 or 1: consecutive): element i at
  /// (i >> shift) * out_stride + (i & (2^shift - 1)) * out_tile_mul + f * out_fdist.  With out_tile_mul = FPW and
  /// out_fdist = 1 a stage writes [element % tile][f] tiles -- the four-step stage A filling an intermediate that is
  /// laid out per stage-B work-group.
  unsigned out_tile_mul;
  /// two-level outer index (0 = unused): outer index o = (o / outer_lo, o % outer_lo); the high part advances by
  /// *_dist_outer_hi, the low part by *_dist_outer.  Lets one launch cover several matrices when the stage's outer
  /// index already has a meaning inside a matrix (second stage of a long column transform of an N-D array).
  long long outer_lo;
  long long in_dist_outer_hi, out_dist_outer_hi;
  int any_order;  // host side only: launch without the in-order barrier (pfa_launch)
  /// byte offsets (from the kernel's dynamic LDS base) of the LDS copy of the leading twiddle tables and of the
  /// store-modifier tables; 0 = the kernel's own layout (behind its image).  Set by launches whose LDS holds more than
  /// one stage configuration (stockham_xcd.hpp).
  unsigned twl_lds_off, stw_lds_off;
  /// 1: blocks b and b + 8 (one XCD, dispatched back to back) take neighbouring groups -- stages whose input segments
  /// are narrower than a 128-byte line (stockham_strided_kernel); the grid is then a multiple of 16
  int pair_xcd;
};


/// Launch-time arguments of the first pass of the two-pass 2-D plan (stockham_rows2d.hpp): `nmat` matrices of
/// n0 rows x N columns (N is the kernel's row length), packed, interleaved.
struct rows2d_args {
  const void* in;  // interleaved complex, or the real plane of the split-storage form
  void* out;
  const void* tw;   // row twiddles (layout: radix_list::tw_off)
  const void* twc;  // W_n0^m, m in [0, n0)
  long long nmat;   // matrices (batch x leading dimensions)
  int n0;           // rows per matrix; n0 % RC == 0
  const void* in_im;  // imaginary planes (split-storage form only)
  void* out_im;
  int any_order;  // host side only: launch without the in-order barrier (pfa_launch)
};

}  // namespace pfa
