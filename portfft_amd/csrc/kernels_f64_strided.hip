// f64 strided work-group kernel instantiations for gfx950 (see kernels_f64.hip for the parameter legend).
#include "kernels_impl.hpp"

namespace pfa {

namespace {
using d = double;
constexpr int NT = 2;
const strided_kernel g_strided_f64[] = {
    make_strided_entry<strided_cfg<d, radix_list<8, 8>, 128, 16, 2, NT>>(),         // 64
    make_strided_entry<strided_cfg<d, radix_list<16, 8>, 128, 16, 2, NT>>(),        // 128
    // with_tin: tiled-input forms for the four-step stage B (lanes element-fastest inside the intermediate's tiles)
    with_tin<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, NT>>(make_strided_entry<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, NT>>()),        // 256
    with_tin<strided_cfg<d, radix_list<8, 8, 8>, 512, 8, 2, NT>>(make_strided_entry<strided_cfg<d, radix_list<8, 8, 8>, 512, 8, 2, NT>>()),       // 512
    with_tin<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, NT>>(make_strided_entry<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, NT>>()),      // 1024
    with_tin<strided_cfg<d, radix_list<16, 16, 8>, 512, 4, 2, NT>>(make_strided_entry<strided_cfg<d, radix_list<16, 16, 8>, 512, 4, 2, NT>>()),     // 2048
    // 16 columns per group (256-byte segments) for stages that are column-shaped on both sides (batch-interleaved
    // layouts, N-D outer dimensions): BI N=256 4.7 -> 5.4 TB/s, N=512 4.4 -> 5.0.  The four-step stages keep the
    // 8-column entries above (their row-shaped side gets worse with more rows per wave: N=65536 2.7 -> 2.35).
    wide(make_strided_entry<strided_cfg<d, radix_list<16, 16>, 256, 16, 2, NT>>()),    // 256
    wide(make_strided_entry<strided_cfg<d, radix_list<8, 8, 8>, 1024, 16, 2, NT>>()),  // 512
};
}  // namespace

const strided_kernel* strided_kernels_f64(int* count) {
  *count = static_cast<int>(sizeof(g_strided_f64) / sizeof(g_strided_f64[0]));
  return g_strided_f64;
}

}  // namespace pfa
