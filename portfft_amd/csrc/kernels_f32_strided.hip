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
  // (SE_FS_B: the four-step stage B of n2 = 1024 -- software-pipelined, lanes element-fastest inside the tiles of the
  //  group-major intermediate: 86 us per 256 MiB chunk against 113-120 for the row-staged 16.8.8 form)
  add_strided_entries<strided_cfg<f, radix_list<32, 32>, 512, 16, 2, NT>, SE_ROWS | SE_PREFETCH | SE_TIN | SE_FS_B>(v, 4);  // 1024
  add_strided_entries<strided_cfg<f, radix_list<16, 16, 8>, 1024, 8, 4, NT>, SE_ROWS | SE_TIN | SE_TIN_W | SE_FS_A | SE_FS_B>(v, 1, 4);  // 2048
  add_strided_entries<strided_cfg<f, radix_list<16, 16, 16>, 1024, 4, 4, NT>, SE_ROWS>(v);     // 4096
  // wide groups (512-byte segments) for stages that are column-shaped on both sides with >= 64 adjacent columns: the
  // second pass of the two-pass 2-D plan (1024 x 1024: n = 128 over 8192 columns) and wide batch-interleaved
  // layouts.  tools/tune_2d.hip: n=128 32 -> 64 columns 5.3-5.5 -> 5.9 TB/s, n=256 32 -> 64 columns 5.5 -> 5.95
  add_strided_entries<strided_cfg<f, radix_list<8, 8>, 512, 64, 2, NT>, SE_WIDE>(v, 2);        // 64
  add_strided_entries<strided_cfg<f, radix_list<8, 16>, 512, 64, 2, NT>, SE_WIDE>(v, 2);       // 128
  add_strided_entries<strided_cfg<f, radix_list<16, 16>, 1024, 64, 2, NT>, SE_WIDE>(v, 1);     // 256
  // n = 1024 with a row-shaped side: 16.8.8 on 1024 lanes stages rows better than the 32.32 prefetch kernel above
  // (four-step N=2^20 2.00 -> 2.15 TB/s, P->BI 3.73 -> 3.89, BI->P 3.99 -> 4.21); column/column stages keep 32.32
  add_strided_entries<strided_cfg<f, radix_list<16, 8, 8>, 1024, 16, 4, NT>, SE_ROWS | SE_ROWISH | SE_TIN | SE_FS_A>(v, 4, 8);  // 1024
  // Four-step stage pairs (tools/tune_fourstep.hip, 256 MiB chunks, writer / reader policies, us per chunk A + B):
  // 16 columns per group (128-byte segments) at two to four work-groups per CU beat 32 columns at one or two --
  // N = 65536: 108 + 90 (16.16 on 512 lanes x 32 columns, row-staged stage B) -> 91 + 82; N = 2^18: 117 + 108
  // (8.8.8 on 1024 lanes x 32 columns) -> 101 + 84.  Only chosen as a pair (strided_kernel::fs_a / fs_b).
  add_strided_entries<strided_cfg<f, radix_list<16, 16>, 256, 16, 2, NT>, SE_TIN | SE_FS_A | SE_FS_B | SE_FS_ONLY>(v, 4);   // 256
  add_strided_entries<strided_cfg<f, radix_list<8, 8, 8>, 512, 16, 2, NT>, SE_TIN | SE_FS_A | SE_FS_B | SE_FS_ONLY>(v, 4);  // 512
  return v;
}
}  // namespace

const strided_kernel* strided_kernels_f32(int* count) {
  static const std::vector<strided_kernel> g = build();
  *count = static_cast<int>(g.size());
  return g.data();
}

}  // namespace pfa
