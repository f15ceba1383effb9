// fp32 kernel instantiations for gfx950.
//   wg_cfg<T, radices, WG, FPW, PADS (padding period in elements), PADW, TWM, OCC, AUX>
// AUX = 2 (nt) on the HBM side: the data is streamed once, measured +4..6 % on the N=4096 shape.
#include "kernels_impl.hpp"

namespace pfa {

namespace {
using f = float;
constexpr int NT = 2;
const spec_kernel g_spec_f32[] = {
    // small lengths: LDS-staged coalesced I/O (STAGED = 1), one or a few lanes per FFT
    make_spec_entry<wg_cfg<f, radix_list<2>, 256, 256, 0, 0, TW_GLOBAL, 4, NT, 1>>(),          // 2
    make_spec_entry<wg_cfg<f, radix_list<4>, 256, 256, 4, 1, TW_GLOBAL, 4, NT, 1>>(),          // 4
    make_spec_entry<wg_cfg<f, radix_list<8>, 256, 256, 8, 1, TW_GLOBAL, 4, NT, 1>>(),          // 8
    make_spec_entry<wg_cfg<f, radix_list<16>, 256, 256, 16, 1, TW_GLOBAL, 4, NT, 1>>(),         // 16
    make_spec_entry<wg_cfg_twl<f, radix_list<8, 4>, 256, 64, 8, 1, 4, NT, 1>>(),        // 32
    make_spec_entry<wg_cfg_twl<f, radix_list<8, 8>, 256, 32, 8, 1, 4, NT, 1>>(),        // 64
    make_spec_entry<wg_cfg_twl<f, radix_list<16, 8>, 256, 32, 16, 1, 4, NT, 1>>(),       // 128
    make_spec_entry<wg_cfg_twl<f, radix_list<16, 16>, 256, 16, 16, 1, 4, NT, 1>>(2),      // 256 (LDS-staged I/O: 6.2 vs 5.6 TB/s -- direct I/O moves 128-byte pieces per FFT here)
    make_spec_entry<wg_cfg<f, radix_list<8, 8, 8>, 256, 4, 16, 1, TW_GLOBAL, 4, NT, 0, 2>>(2),  // 512 (TWL 2: 6.56 vs 6.40 with TW_REGS)
    make_spec_entry<wg_cfg<f, radix_list<16, 8, 8>, 256, 4, 16, 1, TW_GLOBAL, 4, NT, 0, 2>>(2),  // 1024 (TWL 2: tools/tune.hip; 2 groups per work-group: library sweep 5.61 / 5.82 / 5.90 TB/s at 1 / 2 / 4, tuner 6.05-6.34 / 5.99-6.39 / 5.89-5.96)
    make_spec_entry<wg_cfg<f, radix_list<16, 16, 8>, 256, 2, 16, 1, TW_GLOBAL, 4, NT, 0, 2>>(4),  // 2048 (TWL 2: 6.03 vs 5.93; 4 groups per work-group: 5.59 -> 5.82)
    // the headline shape: register-resident twiddles + software-pipelined loads (3 work-groups per CU)
    make_spec_entry_prefetch<wg_cfg<f, radix_list<16, 16, 16>, 256, 1, 16, 1, TW_REGS, 3, NT>>(4),  // 4096
    make_spec_entry<wg_cfg<f, radix_list<32, 16, 16>, 256, 1, 16, 1, TW_REGS, 2, NT>>(4),       // 8192
    // 16384 (128 KiB): register-resident on 512 lanes with 128 VGPRs and a 74 KiB half image -- TWO work-groups per CU cover each
    // other's HBM phases.  tools/tune.hip case 16388, TB/s, persistent grid / one transform per work-group: LDS-resident 32.16.32
    // with its twiddles in registers (one work-group per CU) 5.44 / 4.53, hx 32.32.16 5.78 / **6.11**, 32.16.32 5.63 / 6.11,
    // 256 lanes x 64 values 5.62 / 6.23; through the library 0.62 -> 0.71 of the HBM peak
    make_spec_entry_hx<wg_cfg<f, radix_list<32, 32, 16>, 512, 1, 32, 1, TW_GLOBAL, 4, NT, 0, 1>>(1),  // 16384
    make_spec_entry<wg_cfg<f, radix_list<32, 16, 32>, 512, 1, 0, 0, TW_REGS, 2, NT>>(0),       // 16384 (PFFT_NO_REGRES=1, UNPACKED layouts)
    // beyond the CU's LDS: the transform stays in registers, the exchanges cross a half image (stockham_wg_hx.hpp).
    // tools/tune.hip case 32768, TB/s at a grid of 2 x resident / one transform per work-group: 1024 lanes x 32 points
    // 5.19 / 4.92, 512 lanes x 64 points 4.55 / 4.56, four passes (16.16.16.8, 8.16.16.16) 4.0-4.6; the two-launch
    // four-step plan of this length runs at 3.2
    // (grid: four transforms per work-group -- bench.py g32_15 0.589 with the persistent 2 x resident grid, 0.606 / 0.605 / 0.594 with
    //  4 / 2 / 1 per work-group, profiles/r5_hx_grid_rule.txt; the planned lengths of the band prefer the persistent grid)
    make_spec_entry_hx<wg_cfg<f, radix_list<32, 32, 32>, 1024, 1, 32, 1, TW_GLOBAL, 4, NT, 0, 1>>(4),  // 32768
    // 3 * 2^k and 5 * 2^k families, powers of ten
    make_spec_entry<wg_cfg_twl<f, radix_list<12, 8>, 256, 32, 0, 0, 4, NT, 1>>(),       // 96
    make_spec_entry<wg_cfg_twl<f, radix_list<16, 12>, 256, 16, 16, 1, 4, NT, 1>>(),      // 192 (LDS-staged I/O: 5.96 vs 5.64)
    make_spec_entry<wg_cfg_twl<f, radix_list<8, 8, 6>, 256, 4, 16, 1, 4, NT>>(),         // 384
    make_spec_entry<wg_cfg_twl<f, radix_list<16, 8, 6>, 256, 4, 16, 1, 4, NT>>(),        // 768
    make_spec_entry<wg_cfg_twl<f, radix_list<16, 12, 8>, 256, 2, 16, 1, 4, NT>>(),       // 1536
    make_spec_entry<wg_cfg_twl<f, radix_list<16, 16, 12>, 256, 1, 16, 1, 4, NT>>(),      // 3072
    make_spec_entry<wg_cfg_twl<f, radix_list<24, 16, 16>, 256, 1, 16, 1, 2, NT>>(),      // 6144
    // 12288 (96 KiB): the planner's two-per-CU register-resident plan (choose_hx_params), pre-compiled so that the length keeps its
    // instant commit -- 0.49 -> 0.62 of the HBM peak (tools/perf_hx_pairs.py)
    make_spec_entry_hx<wg_cfg<f, radix_list<32, 24, 16>, 512, 1, 32, 1, TW_GLOBAL, 4, NT, 0, 1>>(1),  // 12288
    make_spec_entry<wg_cfg_twl<f, radix_list<32, 24, 16>, 512, 1, 16, 1, 2, NT>>(),      // 12288 (PFFT_NO_REGRES=1, UNPACKED layouts)
    make_spec_entry<wg_cfg_twl<f, radix_list<10, 8>, 256, 32, 0, 0, 4, NT, 1>>(),       // 80
    make_spec_entry<wg_cfg_twl<f, radix_list<10, 10>, 250, 25, 0, 0, 4, NT, 1>>(),      // 100
    make_spec_entry<wg_cfg_twl<f, radix_list<16, 10>, 256, 16, 0, 0, 4, NT, 1>>(),       // 160
    make_spec_entry<wg_cfg_twl<f, radix_list<8, 8, 5>, 256, 4, 16, 1, 4, NT>>(),         // 320
    make_spec_entry<wg_cfg_twl<f, radix_list<16, 8, 5>, 256, 4, 16, 1, 4, NT>>(),        // 640
    make_spec_entry<wg_cfg_twl<f, radix_list<16, 10, 8>, 256, 2, 16, 1, 4, NT>>(),       // 1280
    make_spec_entry<wg_cfg_twl<f, radix_list<16, 16, 10>, 256, 1, 16, 1, 4, NT>>(),      // 2560
    make_spec_entry<wg_cfg_twl<f, radix_list<20, 16, 16>, 256, 1, 16, 1, 2, NT>>(),      // 5120
    make_spec_entry<wg_cfg_twl<f, radix_list<10, 10, 10>, 200, 2, 0, 0, 4, NT>>(),      // 1000
    make_spec_entry<wg_cfg_twl<f, radix_list<10, 10, 10, 10>, 512, 1, 0, 0, 3, NT>>(),  // 10000
    // cross-lane (in-wave DPP / ds_swizzle transpose) variants of N = R * R: measurement only, chosen with PFFT_XLANE=1
    make_spec_entry_xlane<wg_cfg<f, radix_list<4, 4>, 256, 64, 4, 1, TW_GLOBAL, 4, NT, 1>>(),     // 16
    make_spec_entry_xlane<wg_cfg<f, radix_list<8, 8>, 256, 32, 8, 1, TW_GLOBAL, 4, NT, 1>>(),     // 64
    make_spec_entry_xlane<wg_cfg<f, radix_list<16, 16>, 256, 16, 16, 1, TW_GLOBAL, 4, NT, 1>>(),  // 256
};
}  // namespace

const spec_kernel* spec_kernels_f32(int* count) {
  *count = static_cast<int>(sizeof(g_spec_f32) / sizeof(g_spec_f32[0]));
  return g_spec_f32;
}

hipError_t launch_generic_f32(hipStream_t stream, unsigned grid, size_t lds_bytes, const generic_args& args) {
  bool big = false;
  for (int i = 0; i < args.n_passes; ++i) big = big || args.radix[i] > GENERIC_MAX_SMALL_RADIX;
  if (big) {
    hipLaunchKernelGGL((generic_fft_kernel<float, true>), dim3(grid), dim3(GENERIC_WG), lds_bytes, stream, args);
  } else {
    hipLaunchKernelGGL((generic_fft_kernel<float, false>), dim3(grid), dim3(GENERIC_WG), lds_bytes, stream, args);
  }
  return hipGetLastError();
}

}  // namespace pfa
