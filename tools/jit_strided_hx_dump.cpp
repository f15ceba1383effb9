// prints the register-resident strided plans (strided_hx_candidates) of a list of lengths, and a .hip line per plan that
// tools/kres.py can be fed with (no GPU needed): jit_strided_hx_dump [fpw32 fpw64] n...
#include <cstdio>
#include <cstdlib>
#include "../portfft_amd/csrc/jit.hpp"
int main(int argc, char** argv) {
  for (int prec = 0; prec < 2; ++prec)
    for (int i = 1; i < argc; ++i) {
      const long long n = atoll(argv[i]);
      pfa::wg_params b;
      if (pfa::choose_strided_wide_base(prec, n, prec ? 8 : 16, &b)) {  // groups beyond the LDS: one-per-CU plans
        for (const pfa::wg_params& q : pfa::strided_hx_candidates(b, 160 * 1024, true)) {
          printf("%s n=%-6lld WIDE hx x%d per CU: %s lds=%zu regs=%d\n", prec ? "f64" : "f32", n, q.hx_strided, pfa::wg_cfg_type_name(q).c_str(), pfa::strided_hx_lds_bytes(q), q.regs);
          printf("KRES template __global__ void pfa::stockham_strided_hx_kernel<%s, false, 0, 0>(const pfa::strided_args);\n", pfa::wg_cfg_type_name(q).c_str());
        }
      }
      if (!pfa::choose_strided_params(prec, n, 1 << 20, 160 * 1024, &b, false, prec ? 8 : 16) &&
          !pfa::choose_strided_params(prec, n, 1 << 20, 160 * 1024, &b)) {
        printf("%s n=%lld: no strided plan\n", prec ? "f64" : "f32", n);
        continue;
      }
      printf("%s n=%-6lld LDS-resident %s lds=%zu regs=%d\n", prec ? "f64" : "f32", n, pfa::wg_cfg_type_name(b).c_str(),
             static_cast<size_t>(n) * b.fpw * (prec ? 16 : 8), b.regs);
      for (const pfa::wg_params& q : pfa::strided_hx_candidates(b, 160 * 1024)) {
        printf("   hx x%d per CU: %s lds=%zu regs=%d\n", q.hx_strided, pfa::wg_cfg_type_name(q).c_str(), pfa::strided_hx_lds_bytes(q), q.regs);
        printf("KRES template __global__ void pfa::stockham_strided_hx_kernel<%s, false, 1, 0>(const pfa::strided_args);\n", pfa::wg_cfg_type_name(q).c_str());
        printf("KRES template __global__ void pfa::stockham_strided_hx_kernel<%s, false, 0, 0>(const pfa::strided_args);\n", pfa::wg_cfg_type_name(q).c_str());
      }
    }
}
