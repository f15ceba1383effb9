// fp64 kernel instantiations for gfx950 (see kernels_f32.hip for the parameter legend).
#include "kernels_impl.hpp"

namespace pfa {

namespace {
using d = double;
constexpr int NT = 2;
const spec_kernel g_spec_f64[] = {
    make_spec_entry<wg_cfg<d, radix_list<2>, 256, 256, 0, 0, TW_GLOBAL, 2, NT, 1>>(),          // 2
    make_spec_entry<wg_cfg<d, radix_list<4>, 256, 256, 4, 1, TW_GLOBAL, 2, NT, 1>>(),          // 4
    make_spec_entry<wg_cfg<d, radix_list<8>, 256, 256, 8, 1, TW_GLOBAL, 2, NT, 1>>(),          // 8
    make_spec_entry<wg_cfg<d, radix_list<16>, 256, 128, 16, 1, TW_GLOBAL, 2, NT, 1>>(),         // 16
    make_spec_entry<wg_cfg_twl<d, radix_list<8, 4>, 256, 64, 8, 1, 2, NT, 1>>(),        // 32
    make_spec_entry<wg_cfg_twl<d, radix_list<8, 8>, 256, 32, 8, 1, 2, NT, 1>>(),        // 64
    make_spec_entry<wg_cfg_twl<d, radix_list<16, 8>, 256, 32, 16, 1, 2, NT, 1>>(),       // 128
    make_spec_entry<wg_cfg_twl<d, radix_list<16, 16>, 256, 16, 16, 1, 2, NT>>(),       // 256
    make_spec_entry<wg_cfg_twl<d, radix_list<8, 8, 8>, 256, 4, 16, 1, 2, NT>>(),       // 512
    make_spec_entry<wg_cfg_twl<d, radix_list<16, 8, 8>, 256, 4, 16, 1, 2, NT>>(2),     // 1024 (2 groups per work-group: 5.76 -> 5.93 TB/s)
    make_spec_entry<wg_cfg_twl<d, radix_list<16, 16, 8>, 256, 2, 16, 1, 2, NT>>(2),    // 2048
    make_spec_entry<wg_cfg<d, radix_list<16, 16, 16>, 256, 1, 16, 1, TW_REGS, 1, NT>>(1),      // 4096
    // 8192 (a reference WorkgroupOrGlobal size): register-resident AND software-pipelined -- 32 values per lane leave room for
    // the next transform's 32 in flight (235 VGPRs, no scratch).  tools/tune.hip case 8192064, TB/s at a grid of 2 x resident:
    // LDS-resident 16.8.8.8 5.41, hx 16.16.32 5.40, with the prefetch 5.68, **16.32.16 with the prefetch 6.28**, 32.16.16 5.49-5.76.
    // (the same form for fp32 16384 -- tune case 16387 -- ties the TW_REGS entry: 5.36-5.53 against 5.50)
    // ... and then TWO work-groups of 256 lanes per CU (32 values per lane, 205 VGPRs, 76 KiB of LDS each), no prefetch: tune case
    // 8192065, one transform per work-group: 16.32.16 6.29, 32.16.16 6.10 against 6.13 for the pipelined form on the same box --
    // and through the library on three other boxes 0.75 against 0.66 ... 0.73 (tools/perf_hx_pairs.py): the pair is the entry,
    // the pipelined form (stockham_wg_hx_body PF = 1) stays in the tuner
    make_spec_entry_hx<wg_cfg<d, radix_list<16, 32, 16>, 256, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>(1),  // 8192
    make_spec_entry<wg_cfg_twl<d, radix_list<16, 8, 8, 8>, 512, 1, 16, 1, 2, NT>>(1),   // 8192 (PFFT_NO_REGRES=1, UNPACKED layouts)
    // register-resident form (stockham_wg_hx.hpp; tools/tune.hip case 16384064, TB/s at a grid of 2 x resident): 16.32.32 on
    // 512 lanes 5.93, 16.16.8.8 on 512 / 1024 lanes 5.53 / 5.22, 8.8.16.16 5.53; the four-step plan runs at 3.2
    // (grid: four transforms per work-group -- bench.py g64_14 0.615 persistent, 0.651-0.653 with 8 / 4 / 2 per work-group)
    make_spec_entry_hx<wg_cfg<d, radix_list<16, 32, 32>, 512, 1, 16, 1, TW_GLOBAL, 2, NT, 0, 1>>(4),  // 16384
    make_spec_entry<wg_cfg<d, radix_list<12, 8>, 256, 32, 0, 0, TW_GLOBAL, 2, NT, 1>>(),  // 96 (TWL loses 8 % here)
    make_spec_entry<wg_cfg_twl<d, radix_list<16, 12>, 256, 16, 16, 1, 2, NT>>(),         // 192
    make_spec_entry<wg_cfg_twl<d, radix_list<8, 8, 6>, 256, 4, 16, 1, 2, NT>>(),         // 384
    make_spec_entry<wg_cfg_twl<d, radix_list<16, 8, 6>, 256, 4, 16, 1, 2, NT>>(),        // 768
    make_spec_entry<wg_cfg_twl<d, radix_list<16, 12, 8>, 256, 2, 16, 1, 2, NT>>(),       // 1536
    make_spec_entry<wg_cfg_twl<d, radix_list<16, 16, 12>, 256, 1, 16, 1, 2, NT>>(),      // 3072
    // 6144 (96 KiB): the planner's two-per-CU register-resident plan, pre-compiled -- 0.56 -> 0.72
    make_spec_entry_hx<wg_cfg<d, radix_list<32, 24, 8>, 256, 1, 32, 1, TW_GLOBAL, 2, NT, 0, 1>>(1),  // 6144
    make_spec_entry<wg_cfg_twl<d, radix_list<12, 8, 8, 8>, 768, 1, 0, 0, 2, NT>>(),      // 6144 (PFFT_NO_REGRES=1, UNPACKED layouts)
    make_spec_entry<wg_cfg_twl<d, radix_list<10, 8>, 256, 32, 0, 0, 2, NT, 1>>(),       // 80
    make_spec_entry<wg_cfg_twl<d, radix_list<10, 10>, 250, 25, 0, 0, 2, NT, 1>>(),      // 100
    make_spec_entry<wg_cfg_twl<d, radix_list<10, 10, 10>, 200, 2, 0, 0, 2, NT>>(),      // 1000
    // cross-lane variants (see kernels_f32.hip), chosen with PFFT_XLANE=1
    make_spec_entry_xlane<wg_cfg<d, radix_list<8, 8>, 256, 32, 8, 1, TW_GLOBAL, 2, NT, 1>>(),     // 64
    make_spec_entry_xlane<wg_cfg<d, radix_list<16, 16>, 256, 16, 16, 1, TW_GLOBAL, 2, NT, 1>>(),  // 256
};
}  // namespace

const spec_kernel* spec_kernels_f64(int* count) {
  *count = static_cast<int>(sizeof(g_spec_f64) / sizeof(g_spec_f64[0]));
  return g_spec_f64;
}

hipError_t launch_generic_f64(hipStream_t stream, unsigned grid, size_t lds_bytes, const generic_args& args) {
  bool big = false;
  for (int i = 0; i < args.n_passes; ++i) big = big || args.radix[i] > GENERIC_MAX_SMALL_RADIX;
  if (big) {
    hipLaunchKernelGGL((generic_fft_kernel<double, true>), dim3(grid), dim3(GENERIC_WG), lds_bytes, stream, args);
  } else {
    hipLaunchKernelGGL((generic_fft_kernel<double, false>), dim3(grid), dim3(GENERIC_WG), lds_bytes, stream, args);
  }
  return hipGetLastError();
}

const void* generic_kernel_symbol(int precision, bool big) {
  if (big) {
    return precision == PFFT_PRECISION_F64 ? reinterpret_cast<const void*>(&generic_fft_kernel<double, true>)
                                           : reinterpret_cast<const void*>(&generic_fft_kernel<float, true>);
  }
  return precision == PFFT_PRECISION_F64 ? reinterpret_cast<const void*>(&generic_fft_kernel<double, false>)
                                         : reinterpret_cast<const void*>(&generic_fft_kernel<float, false>);
}

}  // namespace pfa
