// prints the register-resident planner's choice (choose_hx_params) for a list of lengths (no GPU needed)
#include <cstdio>
#include <cstdlib>
#include "../portfft_amd/csrc/jit.hpp"
int main(int argc, char** argv) {
  for (int prec = 0; prec < 2; ++prec)
    for (int i = 1; i < argc; ++i) {
      long long n = atoll(argv[i]);
      pfa::wg_params p;
      if (pfa::choose_hx_params(prec, n, 160 * 1024, &p)) {
        printf("%s hx n=%-6lld %-70s lds=%zu regs=%d occ=%d\n", prec ? "f64" : "f32", n, pfa::wg_cfg_type_name(p).c_str(), pfa::hx_lds_bytes(p), p.regs, p.occ);
      } else printf("%s hx n=%-6lld none\n", prec ? "f64" : "f32", n);
    }
}
