// scratch debug harness for the work-group kernel (small sizes, prints errors)
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>
#include "../portfft_amd/csrc/stockham_wg.hpp"
using namespace pfa;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
template <typename Seq, typename T>
std::vector<cx<T>> make_twiddles() {
  std::vector<cx<T>> tw(Seq::tw_total > 0 ? Seq::tw_total : 1);
  for (int p = 1; p < Seq::count; ++p) {
    const int R = Seq::r[p], Ns = Seq::ns(p);
    for (int t = 1; t < R; ++t) for (int q = 0; q < Ns; ++q) {
      const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)(t * q) / (long double)(Ns * R);
      tw[Seq::tw_off(p) + (t - 1) * Ns + q] = {(T)cosl(a), (T)sinl(a)};
    }
  }
  return tw;
}
template <typename Cfg> void run(const char* name, long long batch, int grid) {
  using T = typename Cfg::T; const int N = Cfg::N;
  std::vector<std::complex<T>> h((size_t)batch * N), o((size_t)batch * N);
  for (size_t i = 0; i < h.size(); ++i) h[i] = {(T)std::sin(0.37 * i + 0.1), (T)std::cos(1.7 * i + 0.3)};
  cx<T>*din, *dout, *dtw; auto tw = make_twiddles<typename Cfg::Seq, T>();
  CK(hipMalloc(&din, h.size() * sizeof(cx<T>))); CK(hipMalloc(&dout, h.size() * sizeof(cx<T>))); CK(hipMalloc(&dtw, tw.size() * sizeof(cx<T>)));
  CK(hipMemcpy(din, h.data(), h.size() * sizeof(cx<T>), hipMemcpyHostToDevice)); CK(hipMemcpy(dtw, tw.data(), tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  CK(hipMemset(dout, 0xff, h.size() * sizeof(cx<T>)));
  auto kern = stockham_wg_kernel<Cfg, false>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, 0, (const cx<T>*)din, dout, (const cx<T>*)dtw, batch, (T)1);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(o.data(), dout, h.size() * sizeof(cx<T>), hipMemcpyDeviceToHost));
  double worst = 0;
  for (long long b = 0; b < batch; ++b) {
    double num = 0, den = 0;
    for (int k = 0; k < N; ++k) {
      std::complex<double> s = 0;
      for (int i = 0; i < N; ++i) s += std::complex<double>(h[b * N + i]) * std::polar(1.0, -2 * M_PI * ((long long)i * k % N) / N);
      num += std::norm(s - std::complex<double>(o[b * N + k])); den += std::norm(s);
      if (b == 0 && k < 4) printf("   k=%d ref=(%g,%g) got=(%g,%g)\n", k, s.real(), s.imag(), (double)o[k].real(), (double)o[k].imag());
    }
    worst = std::max(worst, std::sqrt(num / den));
  }
  printf("%-30s N=%d batch=%lld relL2=%.3e\n", name, N, batch, worst);
}
int main() {
  run<wg_cfg<float, radix_list<16>, 64, 64, 0, 0, TW_GLOBAL, 1>>("r16 single", 130, 2);
  run<wg_cfg<float, radix_list<16, 16>, 64, 4, 0, 0, TW_GLOBAL, 1>>("r16x2 fpw4 nopad", 7, 1);
  run<wg_cfg<float, radix_list<16, 16>, 64, 4, 16, 1, TW_REGS, 1>>("r16x2 fpw4 pad twR", 7, 1);
  run<wg_cfg<float, radix_list<16, 16, 16>, 256, 1, 0, 0, TW_GLOBAL, 1>>("r16x3 nopad", 3, 2);
  run<wg_cfg<float, radix_list<16, 16, 16>, 256, 1, 16, 1, TW_REGS, 4>>("r16x3 pad16 twR", 3, 2);
  run<wg_cfg<float, radix_list<8, 8, 8, 8>, 512, 1, 16, 1, TW_GLOBAL, 1>>("r8x4", 3, 2);
  run<wg_cfg<float, radix_list<4, 3, 5>, 64, 4, 0, 0, TW_GLOBAL, 1>>("r4.3.5 N=60", 9, 2);
  run<wg_cfg<double, radix_list<16, 8>, 64, 2, 0, 0, TW_GLOBAL, 1>>("f64 r16.8", 5, 2);
  return 0;
}
