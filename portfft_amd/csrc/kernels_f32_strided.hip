// f32 strided work-group kernel instantiations for gfx950 (see kernels_f32.hip for the parameter legend).
#include "kernels_impl.hpp"

namespace pfa {

namespace {
using f = float;
constexpr int NT = PFA_AUX_NT;
// strided tier: strided_cfg<T, radices, WG, FPW, OCC, AUX> (TW_GLOBAL + automatic TWL); threads per FFT = WG / FPW.
// Every entry comes with its "writer" and "reader" cache-policy twins (add_strided_entries, strided_kernel::policy).
std::vector<strided_kernel> build() {
  std::vector<strided_kernel> v;
  add_strided_entries<strided_cfg<f, radix_list<8, 8>, 256, 32, 2, NT>, SE_ROWS>(v);           // 64
  add_strided_entries<strided_cfg<f, radix_list<16, 8>, 256, 32, 2, NT>, SE_ROWS>(v);          // 128
  // groups per work-group: tools/perf_gpw.py -- four-step N=65536 x 2Ki 0.799 ms with one, 0.761 ms with four
  add_strided_entries<strided_cfg<f, radix_list<16, 16>, 512, 32, 2, NT>, SE_ROWS>(v, 4);      // 256
  add_strided_entries<strided_cfg<f, radix_list<8, 8, 8>, 1024, 32, 2, NT>, SE_ROWS>(v, 2);    // 512
  add_strided_entries<strided_cfg<f, radix_list<32, 32>, 512, 16, 2, NT>, SE_ROWS | SE_PREFETCH>(v, 4);  // 1024
  add_strided_entries<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, NT>, SE_ROWS>(v);      // 2048
  add_strided_entries<strided_cfg<f, radix_list<16, 16, 16>, 1024, 4, 4, NT>, SE_ROWS>(v);     // 4096
  // wide groups (512-byte segments) for stages that are column-shaped on both sides with >= 64 adjacent columns: the
  // second pass of the two-pass 2-D plan (1024 x 1024: n = 128 over 8192 columns) and wide batch-interleaved
  // layouts.  tools/tune_2d.hip: n=128 32 -> 64 columns 5.3-5.5 -> 5.9 TB/s, n=256 32 -> 64 columns 5.5 -> 5.95
  add_strided_entries<strided_cfg<f, radix_list<8, 8>, 512, 64, 2, NT>, SE_WIDE>(v, 2);        // 64
  add_strided_entries<strided_cfg<f, radix_list<8, 16>, 512, 64, 2, NT>, SE_WIDE>(v, 2);       // 128
  add_strided_entries<strided_cfg<f, radix_list<16, 16>, 1024, 64, 2, NT>, SE_WIDE>(v, 1);     // 256
  // n = 1024 with a row-shaped side: 16.8.8 on 1024 lanes stages rows better than the 32.32 prefetch kernel above
  // (four-step N=2^20 2.00 -> 2.15 TB/s, P->BI 3.73 -> 3.89, BI->P 3.99 -> 4.21); column/column stages keep 32.32
  add_strided_entries<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, NT>, SE_ROWS | SE_ROWISH>(v, 4);  // 1024
  return v;
}
}  // namespace

const strided_kernel* strided_kernels_f32(int* count) {
  static const std::vector<strided_kernel> g = build();
  *count = static_cast<int>(g.size());
  return g.data();
}

}  // namespace pfa
