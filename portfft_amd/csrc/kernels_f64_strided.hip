// f64 strided work-group kernel instantiations for gfx950 (see kernels_f64.hip for the parameter legend).
#include "kernels_impl.hpp"

namespace pfa {

namespace {
using d = double;
constexpr int NT = PFA_AUX_NT;
// Every entry comes with its "writer" and "reader" cache-policy twins (add_strided_entries, strided_kernel::policy).
std::vector<strided_kernel> build() {
  std::vector<strided_kernel> v;
  add_strided_entries<strided_cfg<d, radix_list<8, 8>, 128, 16, 2, NT>>(v);         // 64
  add_strided_entries<strided_cfg<d, radix_list<16, 8>, 128, 16, 2, NT>>(v);        // 128
  // SE_TIN: tiled-input forms for the four-step stage B (lanes element-fastest inside the intermediate's tiles)
  // groups per work-group (last argument): tools/perf_gpw.py, random data -- four-step fp64 N=2^20 x 128 (C3) 1.733 ms
  // with one group per work-group, 1.627 ms with four (the tail of a launch of one-work-group-per-CU kernels);
  // N=2^18 1.500 -> 1.451 ms; N=65536 (n=256) prefers two, N=2^22 (n=2048) one
  add_strided_entries<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, NT>, SE_TIN | SE_FS_A | SE_FS_B | SE_PLAIN_WRITER>(v, 2);     // 256
  add_strided_entries<strided_cfg<d, radix_list<8, 8, 8>, 512, 8, 2, NT>, SE_TIN | SE_FS_A | SE_FS_B | SE_PLAIN_WRITER>(v, 4);    // 512
  // their stage-B partners: software-pipelined tiled-input forms carrying the inter-stage twiddles on their loads (as
  // the n = 1024 entry below) -- tools/tune_fourstep.hip cases 116 / 118: A 100-104 -> 95, B 82 -> 83 / A 103 -> 99, B 88 -> 88
  add_strided_entries<strided_cfg<d, radix_list<16, 16>, 128, 8, 2, NT>, SE_PREFETCH | SE_TIN | SE_FS_B | SE_FS_ONLY | SE_LTW>(v, 2);
  add_strided_entries<strided_cfg<d, radix_list<8, 8, 8>, 512, 8, 2, NT>, SE_PREFETCH | SE_TIN | SE_FS_B | SE_FS_ONLY | SE_LTW>(v, 4);
  add_strided_entries<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, NT>, SE_TIN | SE_FS_A | SE_FS_B | SE_PLAIN_WRITER>(v, 4);   // 1024
  // four-step stage B of n2 = 1024 (C3): the software-pipelined kernel in its tiled-input form -- 86-92 us per 256 MiB
  // chunk against 95-96 (tools/tune_fourstep.hip case 120); stage A keeps the entry above (pipelined: equal)
  // ... and it carries the inter-stage twiddles on its loads (SE_LTW): the modifier is 53 % more VALU instructions on
  // stage A, which has no slack, and is hidden behind stage B's memory time -- A 120 -> 108 us, B 90 -> 91-94
  add_strided_entries<strided_cfg<d, radix_list<16, 8, 8>, 512, 8, 2, NT>, SE_PREFETCH | SE_TIN | SE_FS_B | SE_FS_ONLY | SE_LTW>(v, 4);
  add_strided_entries<strided_cfg<d, radix_list<16, 16, 8>, 512, 4, 2, NT>, SE_TIN | SE_TIN_W | SE_FS_B>(v);  // 2048
  // 16 columns per group (256-byte segments) for stages that are column-shaped on both sides (batch-interleaved
  // layouts, N-D outer dimensions): BI N=256 4.7 -> 5.4 TB/s, N=512 4.4 -> 5.0.  The four-step stages keep the
  // 8-column entries above (their row-shaped side gets worse with more rows per wave: N=65536 2.7 -> 2.35).
  add_strided_entries<strided_cfg<d, radix_list<16, 16>, 256, 16, 2, NT>, SE_WIDE>(v);    // 256
  add_strided_entries<strided_cfg<d, radix_list<8, 8, 8>, 1024, 16, 2, NT>, SE_WIDE>(v);  // 512
  return v;
}
}  // namespace

const strided_kernel* strided_kernels_f64(int* count) {
  static const std::vector<strided_kernel> g = build();
  *count = static_cast<int>(g.size());
  return g.data();
}

}  // namespace pfa
