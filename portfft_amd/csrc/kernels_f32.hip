// fp32 kernel instantiations for gfx950.
//   wg_cfg<T, radices, WG, FPW, PADS, PADW, TWM, OCC, AUX>
// AUX = 2 (nt) on the HBM side: the data is streamed once, measured +4..6 % on the N=4096 shape.
#include "kernels_impl.hpp"

namespace pfa {

namespace {
using f = float;
constexpr int NT = 2;
const spec_kernel g_spec_f32[] = {
    make_spec_entry<wg_cfg<f, radix_list<16, 16>, 256, 16, 4, 1, TW_GLOBAL, 4, NT>>(),         // 256
    make_spec_entry<wg_cfg<f, radix_list<8, 8, 8>, 256, 4, 4, 1, TW_GLOBAL, 4, NT>>(),         // 512
    make_spec_entry<wg_cfg<f, radix_list<16, 8, 8>, 256, 4, 4, 1, TW_GLOBAL, 4, NT>>(),        // 1024
    make_spec_entry<wg_cfg<f, radix_list<16, 16, 8>, 256, 2, 4, 1, TW_GLOBAL, 4, NT>>(),       // 2048
    make_spec_entry<wg_cfg<f, radix_list<16, 16, 16>, 256, 1, 4, 1, TW_REGS, 4, NT>>(),        // 4096
    make_spec_entry<wg_cfg<f, radix_list<32, 16, 16>, 256, 1, 4, 1, TW_GLOBAL, 2, NT>>(),      // 8192
    make_spec_entry<wg_cfg<f, radix_list<32, 32, 16>, 512, 1, 4, 1, TW_GLOBAL, 2, NT>>(),      // 16384
};
// strided tier: wg_cfg<T, radices, WG, FPW, 0, 0, TW_GLOBAL, OCC, AUX>; threads per FFT = WG / FPW
const strided_kernel g_strided_f32[] = {
    make_strided_entry<wg_cfg<f, radix_list<8, 8>, 256, 32, 0, 0, TW_GLOBAL, 2, NT>>(),          // 64
    make_strided_entry<wg_cfg<f, radix_list<16, 8>, 256, 32, 0, 0, TW_GLOBAL, 2, NT>>(),         // 128
    make_strided_entry<wg_cfg<f, radix_list<16, 16>, 256, 16, 0, 0, TW_GLOBAL, 2, NT>>(),        // 256
    make_strided_entry<wg_cfg<f, radix_list<8, 8, 8>, 1024, 16, 0, 0, TW_GLOBAL, 4, NT>>(),      // 512
    make_strided_entry<wg_cfg<f, radix_list<16, 8, 8>, 1024, 16, 0, 0, TW_GLOBAL, 4, NT>>(),     // 1024
    make_strided_entry<wg_cfg<f, radix_list<16, 16, 8>, 1024, 8, 0, 0, TW_GLOBAL, 4, NT>>(),     // 2048
    make_strided_entry<wg_cfg<f, radix_list<16, 16, 16>, 1024, 4, 0, 0, TW_GLOBAL, 4, NT>>(),    // 4096
};
}  // namespace

const strided_kernel* strided_kernels_f32(int* count) {
  *count = static_cast<int>(sizeof(g_strided_f32) / sizeof(g_strided_f32[0]));
  return g_strided_f32;
}

const spec_kernel* spec_kernels_f32(int* count) {
  *count = static_cast<int>(sizeof(g_spec_f32) / sizeof(g_spec_f32[0]));
  return g_spec_f32;
}

hipError_t launch_generic_f32(hipStream_t stream, unsigned grid, size_t lds_bytes, const generic_args& args) {
  hipLaunchKernelGGL(generic_fft_kernel<float>, dim3(grid), dim3(GENERIC_WG), lds_bytes, stream, args);
  return hipGetLastError();
}

}  // namespace pfa
