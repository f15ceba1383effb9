// Instantiations of the first pass of the two-pass 2-D plan (stockham_rows2d.hpp) for gfx950.
//   wg_cfg<T, row radices, WG, RC (rows per work-group = column radix), PADS, PADW, TWM, OCC, AUX, STAGED, TWL>
// Every entry comes with its "writer" cache-policy twin (add_rows2d_entries, rows2d_kernel::policy).
// The last row radix RL must satisfy (n / RL) % WG == 0 (one 2-D butterfly RL x RC per lane and slot).
// Measured on fp32 1024 x 1024 x 256 (tools/tune_2d.hip, profiles/r2_notes.md): RC = 8 with register twiddles,
// two work-groups per CU, 2 groups per work-group: 6.27 TB/s; RC = 4: 5.9-6.07.
#include "kernels_impl.hpp"

namespace pfa {

namespace {
using f = float;
using d = double;
constexpr int NT = PFA_AUX_NT;
std::vector<rows2d_kernel> build() {
  std::vector<rows2d_kernel> v;
  add_rows2d_entries<wg_cfg<f, radix_list<16, 8, 2>, 128, 8, 16, 1, TW_GLOBAL, 4, NT, 0, 2>>(v, 2);   // 256
  add_rows2d_entries<wg_cfg<f, radix_list<16, 8, 4>, 128, 8, 16, 1, TW_GLOBAL, 3, NT, 0, 2>>(v, 2);   // 512
  add_rows2d_entries<wg_cfg<f, radix_list<16, 16, 4>, 256, 8, 16, 1, TW_REGS, 2, NT>>(v, 2);          // 1024
  add_rows2d_entries<wg_cfg<f, radix_list<16, 16, 8>, 256, 4, 16, 1, TW_REGS, 2, NT>>(v, 2);          // 2048
  add_rows2d_entries<wg_cfg<d, radix_list<16, 8, 2>, 128, 4, 16, 1, TW_GLOBAL, 4, NT, 0, 2>>(v, 2);   // 256
  add_rows2d_entries<wg_cfg<d, radix_list<16, 8, 4>, 128, 4, 16, 1, TW_GLOBAL, 2, NT, 0, 2>>(v, 2);   // 512
  add_rows2d_entries<wg_cfg<d, radix_list<16, 16, 4>, 256, 4, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>(v, 1);  // 1024 (TWL 1: 73 KiB, two per CU: 6.2 TB/s; TWL 2 = 86 KiB, one per CU: 5.0)
  add_rows2d_entries<wg_cfg<d, radix_list<16, 16, 8>, 256, 2, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>(v, 1);  // 2048 (TWL 1: two work-groups per CU)
  return v;
}
}  // namespace

const rows2d_kernel* rows2d_kernels(int* count) {
  static const std::vector<rows2d_kernel> g = build();
  *count = static_cast<int>(g.size());
  return g.data();
}

}  // namespace pfa
