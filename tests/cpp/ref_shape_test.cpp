// User code written against include/portfft/portfft.hpp in the CALL SHAPES of the reference's own two callers:
//   * the unit-test driver (test/unit_test/fft_test_utils.hpp:286-333): host -> device copies that return events, a
//     `dependencies` vector handed to compute_forward/backward (all eight USM overloads), the returned event handed
//     to the copy back, queue.wait();
//   * the bench loop (test/bench/portfft/launch_bench.hpp:118-139): `runs` chained submissions, each one depending on
//     the event of the previous one, a single wait() on the last event, host timing around the chain.
// Plus what those shapes rely on: an event completes with ITS submission (not with the whole stream), dependencies
// order work across streams, and copies of a committed_descriptor own their scratch
// (committed_descriptor_impl.hpp:774-817).  Written from scratch; only the argument order and the control flow of the
// callers are mirrored.
//   hipcc -std=c++17 -I include tests/cpp/ref_shape_test.cpp -L portfft_amd -lportfft_amd -o build/ref_shape_test
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <memory>
#include <vector>

#include <portfft/portfft.hpp>

#define REQUIRE(c)                                               \
  do {                                                           \
    if (!(c)) {                                                  \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                  \
    }                                                            \
  } while (0)
#define HIP_OK(x) REQUIRE((x) == hipSuccess)

namespace {

template <typename T>
std::shared_ptr<T> make_device(std::size_t count) {
  void* p = nullptr;
  if (hipMalloc(&p, count * sizeof(T)) != hipSuccess) return nullptr;
  return std::shared_ptr<T>(static_cast<T*>(p), [](T* q) { (void)hipFree(q); });
}

template <typename T>
std::vector<std::complex<T>> make_input(std::size_t n, unsigned seed) {
  std::vector<std::complex<T>> v(n);
  unsigned s = seed;
  for (auto& e : v) {
    s = s * 1664525u + 1013904223u;
    const T re = static_cast<T>((s >> 8) & 0xFFFF) / T(32768) - T(1);
    s = s * 1664525u + 1013904223u;
    const T im = static_cast<T>((s >> 8) & 0xFFFF) / T(32768) - T(1);
    e = {re, im};
  }
  return v;
}

/// forward DFT of every transform in double precision (direct sum for short lengths, recursive radix-2 otherwise)
void fft_rec(std::vector<std::complex<double>>& x) {
  const std::size_t n = x.size();
  if (n == 1) return;
  if (n % 2 != 0) {
    std::vector<std::complex<double>> y(n);
    for (std::size_t k = 0; k < n; ++k) {
      std::complex<double> acc = 0;
      for (std::size_t j = 0; j < n; ++j) {
        const double a = -2.0 * M_PI * static_cast<double>((j * k) % n) / static_cast<double>(n);
        acc += x[j] * std::complex<double>(std::cos(a), std::sin(a));
      }
      y[k] = acc;
    }
    x = y;
    return;
  }
  std::vector<std::complex<double>> e(n / 2), o(n / 2);
  for (std::size_t i = 0; i < n / 2; ++i) {
    e[i] = x[2 * i];
    o[i] = x[2 * i + 1];
  }
  fft_rec(e);
  fft_rec(o);
  for (std::size_t k = 0; k < n / 2; ++k) {
    const double a = -2.0 * M_PI * static_cast<double>(k) / static_cast<double>(n);
    const std::complex<double> t = o[k] * std::complex<double>(std::cos(a), std::sin(a));
    x[k] = e[k] + t;
    x[k + n / 2] = e[k] - t;
  }
}

template <typename T>
double rel_l2(const std::vector<std::complex<T>>& got, const std::vector<std::complex<T>>& in, std::size_t n,
              std::size_t batch, bool backward) {
  double num = 0, den = 0;
  for (std::size_t b = 0; b < batch; ++b) {
    std::vector<std::complex<double>> x(n);
    for (std::size_t i = 0; i < n; ++i) {
      const auto v = in[b * n + i];
      x[i] = backward ? std::complex<double>(v.real(), -v.imag()) : std::complex<double>(v.real(), v.imag());
    }
    fft_rec(x);
    for (std::size_t i = 0; i < n; ++i) {
      const std::complex<double> r = backward ? std::conj(x[i]) : x[i];
      const std::complex<double> g(got[b * n + i].real(), got[b * n + i].imag());
      num += std::norm(g - r);
      den += std::norm(r);
    }
  }
  return std::sqrt(num / den);
}

/// The unit-test driver's shape: copies with events, `dependencies`, compute, dependent copy back, queue.wait().
template <typename T, portfft::direction Dir, portfft::complex_storage Storage, bool OutOfPlace>
int test_driver_shape(std::size_t n, std::size_t batch, double tol) {
  using namespace portfft;
  hipStream_t stream = nullptr;
  HIP_OK(hipStreamCreate(&stream));
  queue q(stream);
  descriptor<T, domain::COMPLEX> desc({n});
  desc.number_of_transforms = batch;
  desc.complex_storage = Storage;
  desc.placement = OutOfPlace ? placement::OUT_OF_PLACE : placement::IN_PLACE;
  auto committed_descriptor = desc.commit(q);

  const auto host_input = make_input<T>(n * batch, 7u + static_cast<unsigned>(n));
  std::vector<std::complex<T>> host_output(n * batch);
  std::vector<T> host_input_real(n * batch), host_input_imag(n * batch), host_output_real(n * batch),
      host_output_imag(n * batch);
  for (std::size_t i = 0; i < n * batch; ++i) {
    host_input_real[i] = host_input[i].real();
    host_input_imag[i] = host_input[i].imag();
  }
  auto device_input = make_device<std::complex<T>>(n * batch);
  auto device_output = make_device<std::complex<T>>(n * batch);
  auto device_input_imag = make_device<T>(n * batch);
  auto device_output_imag = make_device<T>(n * batch);
  REQUIRE(device_input && device_output && device_input_imag && device_output_imag);

  std::vector<event> dependencies;
  if constexpr (Storage == complex_storage::INTERLEAVED_COMPLEX) {
    dependencies.push_back(q.copy(host_input.data(), device_input.get(), n * batch));
  } else {
    dependencies.push_back(q.copy(host_input_real.data(), reinterpret_cast<T*>(device_input.get()), n * batch));
    dependencies.push_back(q.copy(host_input_imag.data(), device_input_imag.get(), n * batch));
  }
  T* const in_real = reinterpret_cast<T*>(device_input.get());
  T* const out_real = reinterpret_cast<T*>(device_output.get());

  event fft_event = [&]() {
    if constexpr (OutOfPlace) {
      if constexpr (Dir == direction::FORWARD) {
        if constexpr (Storage == complex_storage::INTERLEAVED_COMPLEX) {
          return committed_descriptor.compute_forward(device_input.get(), device_output.get(), dependencies);
        } else {
          return committed_descriptor.compute_forward(in_real, device_input_imag.get(), out_real,
                                                      device_output_imag.get(), dependencies);
        }
      } else {
        if constexpr (Storage == complex_storage::INTERLEAVED_COMPLEX) {
          return committed_descriptor.compute_backward(device_input.get(), device_output.get(), dependencies);
        } else {
          return committed_descriptor.compute_backward(in_real, device_input_imag.get(), out_real,
                                                       device_output_imag.get(), dependencies);
        }
      }
    } else {
      if constexpr (Dir == direction::FORWARD) {
        if constexpr (Storage == complex_storage::INTERLEAVED_COMPLEX) {
          return committed_descriptor.compute_forward(device_input.get(), dependencies);
        } else {
          return committed_descriptor.compute_forward(in_real, device_input_imag.get(), dependencies);
        }
      } else {
        if constexpr (Storage == complex_storage::INTERLEAVED_COMPLEX) {
          return committed_descriptor.compute_backward(device_input.get(), dependencies);
        } else {
          return committed_descriptor.compute_backward(in_real, device_input_imag.get(), dependencies);
        }
      }
    }
  }();

  if constexpr (Storage == complex_storage::INTERLEAVED_COMPLEX) {
    q.copy(OutOfPlace ? device_output.get() : device_input.get(), host_output.data(), n * batch, {fft_event});
  } else {
    q.copy(OutOfPlace ? out_real : in_real, host_output_real.data(), n * batch, {fft_event});
    q.copy(OutOfPlace ? device_output_imag.get() : device_input_imag.get(), host_output_imag.data(), n * batch,
           {fft_event});
  }
  q.wait_and_throw();
  if constexpr (Storage == complex_storage::SPLIT_COMPLEX) {
    for (std::size_t i = 0; i < n * batch; ++i) host_output[i] = {host_output_real[i], host_output_imag[i]};
  }
  const double err = rel_l2(host_output, host_input, n, batch, Dir == direction::BACKWARD);
  std::printf("driver shape n=%zu batch=%zu %s %s %s rel-L2 %.2e\n", n, batch, sizeof(T) == 4 ? "f32" : "f64",
              Dir == direction::FORWARD ? "fwd" : "bwd", OutOfPlace ? "oop" : "ip", err);
  REQUIRE(err < tol);
  HIP_OK(hipStreamDestroy(stream));
  return 0;
}

/// The bench loop's shape: `runs` chained submissions through `dependencies`, one wait on the last event.
int test_bench_loop_shape() {
  using namespace portfft;
  const std::size_t n = 4096, batch = 2048, runs = 10, num_inputs = 2;
  hipStream_t stream = nullptr;
  HIP_OK(hipStreamCreate(&stream));
  queue q(stream);
  descriptor<float, domain::COMPLEX> desc({n});
  desc.number_of_transforms = batch;
  auto committed = desc.commit(q);
  const std::size_t num_elements = n * batch;
  const auto host_forward_data = make_input<float>(num_elements, 99u);
  std::vector<std::shared_ptr<std::complex<float>>> device_inputs;
  for (std::size_t i = 0; i < num_inputs; ++i) device_inputs.push_back(make_device<std::complex<float>>(num_elements));
  auto out_dev = make_device<std::complex<float>>(num_elements);
  REQUIRE(device_inputs[0] && device_inputs[1] && out_dev);

  std::vector<event> dependencies;
  dependencies.reserve(1);
  double elapsed_sum = 0;
  for (int iteration = 0; iteration < 3; ++iteration) {
    dependencies.clear();
    for (auto& in_dev : device_inputs) q.copy(host_forward_data.data(), in_dev.get(), num_elements);
    q.wait_and_throw();
    const auto start = std::chrono::high_resolution_clock::now();
    dependencies.emplace_back(committed.compute_forward(device_inputs[0].get(), out_dev.get()));
    for (std::size_t r = 1; r != runs; r += 1) {
      dependencies[0] = committed.compute_forward(device_inputs[r % num_inputs].get(), out_dev.get(), dependencies);
    }
    dependencies[0].wait();
    const auto end = std::chrono::high_resolution_clock::now();
    REQUIRE(dependencies[0].is_complete());
    elapsed_sum += std::chrono::duration<double>(end - start).count() / static_cast<double>(runs);
  }
  std::vector<std::complex<float>> host_output(num_elements);
  q.copy(out_dev.get(), host_output.data(), num_elements).wait();
  const std::vector<std::complex<float>> first(host_forward_data.begin(), host_forward_data.begin() + 4 * n);
  const std::vector<std::complex<float>> got(host_output.begin(), host_output.begin() + 4 * n);
  const double err = rel_l2(got, first, n, 4, false);
  std::printf("bench-loop shape: %.1f us per submission, rel-L2 %.2e\n", elapsed_sum / 3 * 1e6, err);
  REQUIRE(err < 2e-6);
  HIP_OK(hipStreamDestroy(stream));
  return 0;
}

/// An event completes with its own submission; a dependency orders work across streams.
int test_event_semantics() {
  using namespace portfft;
  const std::size_t n = 4096, batch = 16384;  // ~0.2 ms per transform launch
  hipStream_t s_fft = nullptr, s_other = nullptr;
  HIP_OK(hipStreamCreate(&s_fft));
  HIP_OK(hipStreamCreate(&s_other));
  queue q(s_fft), q_other(s_other);
  descriptor<float, domain::COMPLEX> desc({n});
  desc.number_of_transforms = batch;
  auto plan = desc.commit(q);
  const auto host = make_input<float>(n * batch, 5u);
  auto a = make_device<std::complex<float>>(n * batch);
  auto b = make_device<std::complex<float>>(n * batch);
  REQUIRE(a && b);
  // the input is produced on ANOTHER stream; only the dependency orders the transform behind it
  HIP_OK(hipMemsetAsync(a.get(), 0, n * batch * sizeof(std::complex<float>), s_other));
  event produced = q_other.copy(host.data(), a.get(), n * batch);
  event first = plan.compute_forward(a.get(), b.get(), {produced});
  // queue 20 more submissions behind the first: its event must complete long before the stream drains
  event last;
  for (int i = 0; i < 20; ++i) last = plan.compute_forward(a.get(), b.get());
  first.wait();
  REQUIRE(first.is_complete());
  const bool last_done_early = last.is_complete();
  last.wait();
  REQUIRE(last.is_complete());
  std::printf("event semantics: first submission complete while the last one was %s\n",
              last_done_early ? "already done (fast device; not conclusive)" : "still pending");
  std::vector<std::complex<float>> out(4 * n);
  q.copy(b.get(), out.data(), 4 * n, {last}).wait();
  const std::vector<std::complex<float>> in4(host.begin(), host.begin() + 4 * n);
  REQUIRE(rel_l2(out, in4, n, 4, false) < 2e-6);
  REQUIRE(event().is_complete());  // default-constructed: nothing to wait for
  event().wait();
  HIP_OK(hipStreamDestroy(s_fft));
  HIP_OK(hipStreamDestroy(s_other));
  return 0;
}

/// Copies share kernels and twiddles and own their scratch; a copy outlives the original.
int test_copy_semantics() {
  using namespace portfft;
  const std::size_t n = 65536, batch = 8;  // GLOBAL tier: uses scratch
  queue q;
  descriptor<float, domain::COMPLEX> desc({n});
  desc.number_of_transforms = batch;
  const auto host = make_input<float>(n * batch, 11u);
  auto in = make_device<std::complex<float>>(n * batch);
  auto out1 = make_device<std::complex<float>>(n * batch);
  auto out2 = make_device<std::complex<float>>(n * batch);
  REQUIRE(in && out1 && out2);
  q.copy(host.data(), in.get(), n * batch).wait();
  std::unique_ptr<committed_descriptor<float, domain::COMPLEX>> copy;
  {
    auto original = desc.commit(q);
    REQUIRE(original.info().scratch_bytes >= n * batch * sizeof(std::complex<float>));
    copy = std::make_unique<committed_descriptor<float, domain::COMPLEX>>(original);  // copy constructor
    auto assigned = desc.commit(q);
    assigned = original;  // copy assignment
    original.compute_forward(in.get(), out1.get());
    assigned.compute_forward(in.get(), out2.get()).wait();
  }  // the original and `assigned` are destroyed here; the copy keeps the shared twiddles alive
  std::vector<std::complex<float>> h1(n * batch), h2(n * batch), h3(n * batch);
  q.copy(out1.get(), h1.data(), n * batch);
  q.copy(out2.get(), h2.data(), n * batch).wait();
  copy->compute_forward(in.get(), out1.get()).wait();
  q.copy(out1.get(), h3.data(), n * batch).wait();
  REQUIRE(h1 == h2 && h1 == h3);  // same kernels, same tables: bit-identical
  REQUIRE(rel_l2(h1, host, n, batch, false) < 2e-6);
  std::printf("copy semantics OK\n");
  return 0;
}

}  // namespace

int main() {
  using portfft::complex_storage;
  using portfft::direction;
  int rc = 0;
  try {
    rc |= test_driver_shape<float, direction::FORWARD, complex_storage::INTERLEAVED_COMPLEX, true>(4096, 8, 2e-6);
    rc |= test_driver_shape<float, direction::BACKWARD, complex_storage::INTERLEAVED_COMPLEX, true>(4096, 8, 2e-6);
    rc |= test_driver_shape<float, direction::FORWARD, complex_storage::INTERLEAVED_COMPLEX, false>(64, 1, 2e-6);
    rc |= test_driver_shape<float, direction::BACKWARD, complex_storage::INTERLEAVED_COMPLEX, false>(1536, 3, 2e-6);
    rc |= test_driver_shape<float, direction::FORWARD, complex_storage::SPLIT_COMPLEX, true>(1024, 5, 2e-6);
    rc |= test_driver_shape<float, direction::BACKWARD, complex_storage::SPLIT_COMPLEX, true>(1024, 5, 2e-6);
    rc |= test_driver_shape<float, direction::FORWARD, complex_storage::SPLIT_COMPLEX, false>(256, 33, 2e-6);
    rc |= test_driver_shape<float, direction::BACKWARD, complex_storage::SPLIT_COMPLEX, false>(256, 33, 2e-6);
    rc |= test_driver_shape<double, direction::FORWARD, complex_storage::INTERLEAVED_COMPLEX, true>(65536, 2, 5e-15);
    rc |= test_driver_shape<double, direction::BACKWARD, complex_storage::SPLIT_COMPLEX, false>(512, 4, 5e-15);
    rc |= test_bench_loop_shape();
    rc |= test_event_semantics();
    rc |= test_copy_semantics();
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 1;
  }
  if (rc == 0) std::printf("ref shape OK\n");
  return rc;
}
