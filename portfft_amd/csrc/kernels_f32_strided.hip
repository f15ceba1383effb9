// f32 strided work-group kernel instantiations for gfx950 (see kernels_f32.hip for the parameter legend).
#include "kernels_impl.hpp"

namespace pfa {

namespace {
using f = float;
constexpr int NT = 2;
// strided tier: strided_cfg<T, radices, WG, FPW, OCC, AUX> (TW_GLOBAL + automatic TWL); threads per FFT = WG / FPW
const strided_kernel g_strided_f32[] = {
    with_rows<strided_cfg<f, radix_list<8, 8>, 256, 32, 2, NT>>(make_strided_entry<strided_cfg<f, radix_list<8, 8>, 256, 32, 2, NT>>()),          // 64
    with_rows<strided_cfg<f, radix_list<16, 8>, 256, 32, 2, NT>>(make_strided_entry<strided_cfg<f, radix_list<16, 8>, 256, 32, 2, NT>>()),         // 128
    with_rows<strided_cfg<f, radix_list<16, 16>, 512, 32, 2, NT>>(make_strided_entry<strided_cfg<f, radix_list<16, 16>, 512, 32, 2, NT>>()),        // 256
    with_rows<strided_cfg<f, radix_list<8, 8, 8>, 1024, 32, 2, NT>>(make_strided_entry<strided_cfg<f, radix_list<8, 8, 8>, 1024, 32, 2, NT>>()),      // 512
    with_rows<strided_cfg<f, radix_list<32, 32>, 512, 16, 2, NT>>(make_strided_entry_prefetch<strided_cfg<f, radix_list<32, 32>, 512, 16, 2, NT>>(4)),  // 1024
    with_rows<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, NT>>(make_strided_entry<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, NT>>()),     // 2048
    with_rows<strided_cfg<f, radix_list<16, 16, 16>, 1024, 4, 4, NT>>(make_strided_entry<strided_cfg<f, radix_list<16, 16, 16>, 1024, 4, 4, NT>>()),    // 4096
    // wide groups (512-byte segments) for stages that are column-shaped on both sides with >= 64 adjacent columns: the
    // second pass of the two-pass 2-D plan (1024 x 1024: n = 128 over 8192 columns) and wide batch-interleaved
    // layouts.  tools/tune_2d.hip: n=128 32 -> 64 columns 5.3-5.5 -> 5.9 TB/s, n=256 32 -> 64 columns 5.5 -> 5.95
    wide(make_strided_entry<strided_cfg<f, radix_list<8, 8>, 512, 64, 2, NT>>(2)),        // 64
    wide(make_strided_entry<strided_cfg<f, radix_list<8, 16>, 512, 64, 2, NT>>(2)),       // 128
    wide(make_strided_entry<strided_cfg<f, radix_list<16, 16>, 1024, 64, 2, NT>>(1)),     // 256
    // n = 1024 with a row-shaped side: 16.8.8 on 1024 lanes stages rows better than the 32.32 prefetch kernel above
    // (four-step N=2^20 2.00 -> 2.15 TB/s, P->BI 3.73 -> 3.89, BI->P 3.99 -> 4.21); column/column stages keep 32.32
    rowish(with_rows<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, NT>>(
        make_strided_entry<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, NT>>())),  // 1024
};
}  // namespace

const strided_kernel* strided_kernels_f32(int* count) {
  *count = static_cast<int>(sizeof(g_strided_f32) / sizeof(g_strided_f32[0]));
  return g_strided_f32;
}

}  // namespace pfa
