// C++ user-code test of the facade: reads like the reference's own usage (README.md, test/unit_test/fft_test_utils.hpp):
// descriptor -> commit -> compute_forward -> wait, exception types, getters.
//   hipcc -std=c++17 -I include tests/cpp/facade_test.cpp -L portfft_amd -lportfft_amd -o build/facade_test
// With argument "host" only the host-side checks run (no GPU needed).
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <vector>

#include <portfft/portfft.hpp>

#define REQUIRE(c)                                               \
  do {                                                           \
    if (!(c)) {                                                  \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                  \
    }                                                            \
  } while (0)

int host_checks() {
  using namespace portfft;
  descriptor<float, domain::COMPLEX> desc({2, 3});
  REQUIRE(desc.forward_strides == (std::vector<std::size_t>{3, 1}));
  REQUIRE(desc.forward_distance == 6 && desc.backward_distance == 6);
  desc.number_of_transforms = 2;
  desc.forward_strides = {8, 3};
  desc.backward_strides = {2, 4};
  desc.forward_distance = 15;
  desc.backward_distance = 1;
  desc.forward_offset = 3;
  desc.backward_offset = 5;
  REQUIRE(desc.get_input_count(direction::FORWARD) == 33);   // test/unit_test/descriptor.cpp:107
  REQUIRE(desc.get_output_count(direction::FORWARD) == 17);  // test/unit_test/descriptor.cpp:108
  queue q;
  bool threw = false;
  try {
    descriptor<float, domain::COMPLEX> bad({0});
    bad.commit(q);
  } catch (const invalid_configuration&) {
    threw = true;
  }
  REQUIRE(threw);
  threw = false;
  try {
    descriptor<float, domain::REAL> real({8});
    real.commit(q);
  } catch (const unsupported_configuration&) {
    threw = true;
  }
  REQUIRE(threw);
  std::printf("host checks OK\n");
  return 0;
}

template <typename T>
int device_checks(std::size_t n, std::size_t batch, double tol) {
  using namespace portfft;
  using cplx = std::complex<T>;
  std::vector<cplx> h(n * batch), r(n * batch);
  for (std::size_t i = 0; i < h.size(); ++i) h[i] = cplx(std::sin(0.37 * i + 0.1), std::cos(1.7 * i + 0.3));
  cplx *din, *dout;
  REQUIRE(hipMalloc(&din, h.size() * sizeof(cplx)) == hipSuccess);
  REQUIRE(hipMalloc(&dout, h.size() * sizeof(cplx)) == hipSuccess);
  REQUIRE(hipMemcpy(din, h.data(), h.size() * sizeof(cplx), hipMemcpyHostToDevice) == hipSuccess);
  hipStream_t stream;
  REQUIRE(hipStreamCreate(&stream) == hipSuccess);
  queue q(stream);
  descriptor<T, domain::COMPLEX> desc({n});
  desc.number_of_transforms = batch;
  auto committed = desc.commit(q);
  committed.compute_forward(din, dout).wait();
  REQUIRE(hipMemcpy(r.data(), dout, r.size() * sizeof(cplx), hipMemcpyDeviceToHost) == hipSuccess);
  double worst = 0;
  for (std::size_t b = 0; b < batch; ++b) {
    double num = 0, den = 0;
    for (std::size_t k = 0; k < n; ++k) {
      std::complex<double> s = 0;
      for (std::size_t i = 0; i < n; ++i) {
        s += std::complex<double>(h[b * n + i]) * std::polar(1.0, -2 * M_PI * double((i * k) % n) / double(n));
      }
      num += std::norm(s - std::complex<double>(r[b * n + k]));
      den += std::norm(s);
    }
    worst = std::max(worst, std::sqrt(num / den));
  }
  std::printf("N=%zu batch=%zu %s forward rel-L2 %.3e\n", n, batch, sizeof(T) == 4 ? "f32" : "f64", worst);
  REQUIRE(worst < tol);
  // in-place backward of the result returns N * input
  committed.compute_backward(dout).wait();
  REQUIRE(hipMemcpy(r.data(), dout, r.size() * sizeof(cplx), hipMemcpyDeviceToHost) == hipSuccess);
  double num = 0, den = 0;
  for (std::size_t i = 0; i < h.size(); ++i) {
    num += std::norm(std::complex<double>(r[i]) / double(n) - std::complex<double>(h[i]));
    den += std::norm(std::complex<double>(h[i]));
  }
  REQUIRE(std::sqrt(num / den) < tol);
  bool threw = false;
  try {
    committed.compute_forward(reinterpret_cast<T*>(din), reinterpret_cast<T*>(dout));  // split call, interleaved plan
  } catch (const invalid_configuration&) {
    threw = true;
  }
  REQUIRE(threw);
  (void)hipFree(din);
  (void)hipFree(dout);
  (void)hipStreamDestroy(stream);
  return 0;
}

int main(int argc, char** argv) {
  if (host_checks() != 0) return 1;
  if (argc > 1 && std::strcmp(argv[1], "host") == 0) return 0;
  if (device_checks<float>(64, 1, 1e-6) != 0) return 1;
  if (device_checks<float>(4096, 3, 2e-6) != 0) return 1;
  if (device_checks<double>(1024, 2, 1e-14) != 0) return 1;
  std::printf("facade OK\n");
  return 0;
}
