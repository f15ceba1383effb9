// One process, one host thread per device ("per-GPU plans driven by their own host thread"; the
// reference ties a committed_descriptor to ONE queue, committed_descriptor_impl.hpp:108-111, and leaves the rest to
// its caller).  Every thread commits -- concurrently, behind a start barrier -- the headline configuration at
// batch / G, a runtime-specialised length (hiprtc + the shared on-disk cache: cold when PFFT_JIT_CACHE_DIR points at an
// empty directory) and the XCD-local four-step plan, then the threads execute between barriers and check sampled
// transforms against a double-precision DFT on the host.  On a 1-GPU box: 4 threads x 4 streams on device 0.
// Also without a device ("host"): the threads race through the host-only entry points and the runtime compiler
// (what the ThreadSanitizer build runs, tools/sanitize_host.sh tsan).
//   hipcc -std=c++17 -pthread -I include tests/cpp/multi_device_test.cpp -L portfft_amd -lportfft_amd -o build/multi_device_test
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <complex>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <portfft/portfft.hpp>

#include "../../portfft_amd/csrc/jit.hpp"

namespace {

class barrier_t {  // (std::barrier is C++20)
 public:
  explicit barrier_t(int n) : n_(n) {}
  void arrive_and_wait() {
    std::unique_lock<std::mutex> lock(m_);
    const int gen = gen_;
    if (++count_ == n_) {
      count_ = 0;
      ++gen_;
      cv_.notify_all();
    } else {
      cv_.wait(lock, [&] { return gen != gen_; });
    }
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  int n_, count_ = 0, gen_ = 0;
};

std::atomic<int> g_fail{0};
#define CHECK(c, ...)                                   \
  do {                                                  \
    if (!(c)) {                                         \
      ++g_fail;                                         \
      std::printf("FAILED %s:%d: %s  ", __FILE__, __LINE__, #c); \
      std::printf(__VA_ARGS__);                         \
      std::printf("\n");                                \
    }                                                   \
  } while (0)

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
    const std::complex<double> w = std::complex<double>(std::cos(a), std::sin(a)) * o[k];
    x[k] = e[k] + w;
    x[k + n / 2] = e[k] - w;
  }
}

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + seed * 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 31;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 29;
    p[i] = static_cast<float>(static_cast<double>(z >> 11) * (2.0 / 9007199254740992.0) - 1.0);
  }
}

/// (experiments: MDT_XCD_BATCH, MDT_THREADS)
std::size_t env_or(const char* name, std::size_t dflt) {
  const char* e = std::getenv(name);
  return e != nullptr ? static_cast<std::size_t>(std::atoll(e)) : dflt;
}

struct job {
  std::size_t n, batch;
  const char* what;
};

/// one thread = one device (or one stream of device 0): commit three plans, execute them, check, report
void worker(int tid, int device, int n_threads, barrier_t* bar, std::vector<double>* ms_out) {
  if (hipSetDevice(device) != hipSuccess) {
    ++g_fail;
    return;
  }
  hipStream_t stream = nullptr;
  CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess, "stream");
  portfft::queue q(stream);
  const job jobs[3] = {{4096, static_cast<std::size_t>(65536 / n_threads), "headline N=4096"},
                       {3000, 1024, "runtime-specialised N=3000"},
                       {std::size_t{1} << 18, env_or("MDT_XCD_BATCH", 256), "XCD-local N=2^18"}};
  std::vector<portfft::committed_descriptor<float, portfft::domain::COMPLEX>> plans;
  bar->arrive_and_wait();  // every thread commits at the same time: hiprtc, the on-disk cache, the XCD census
  try {
    for (const job& j : jobs) {
      portfft::descriptor<float, portfft::domain::COMPLEX> d({j.n});
      d.number_of_transforms = j.batch;
      plans.push_back(d.commit(q));
    }
  } catch (const std::exception& e) {
    CHECK(false, "commit on thread %d threw: %s", tid, e.what());
    return;
  }
  for (int k = 0; k < 3; ++k) {
    const job& j = jobs[k];
    const std::size_t count = j.n * j.batch;
    std::complex<float>*in = nullptr, *out = nullptr;
    CHECK(hipMalloc(&in, count * 8) == hipSuccess && hipMalloc(&out, count * 8) == hipSuccess, "hipMalloc");
    fill_kernel<<<1024, 256, 0, stream>>>(reinterpret_cast<float*>(in), 2 * count, 100u * tid + k);
    CHECK(hipStreamSynchronize(stream) == hipSuccess, "fill");
    plans[k].compute_forward(in, out).wait();  // warm-up
    bar->arrive_and_wait();                    // all threads launch together: the devices (or streams) run side by side
    const auto t0 = std::chrono::steady_clock::now();
    const int reps = 10;
    portfft::event ev;
    for (int r = 0; r < reps; ++r) ev = plans[k].compute_forward(in, out);
    ev.wait();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    (*ms_out)[static_cast<std::size_t>(tid) * 3 + k] = ms;
    bar->arrive_and_wait();
    // sampled transforms against a double-precision DFT
    for (std::size_t b : {std::size_t{0}, j.batch / 2, j.batch - 1}) {
      std::vector<std::complex<float>> hx(j.n), hy(j.n);
      CHECK(hipMemcpy(hx.data(), in + b * j.n, j.n * 8, hipMemcpyDeviceToHost) == hipSuccess, "copy");
      CHECK(hipMemcpy(hy.data(), out + b * j.n, j.n * 8, hipMemcpyDeviceToHost) == hipSuccess, "copy");
      std::vector<std::complex<double>> ref(hx.begin(), hx.end());
      fft_rec(ref);
      double num = 0, den = 0;
      for (std::size_t i = 0; i < j.n; ++i) {
        num += std::norm(std::complex<double>(hy[i]) - ref[i]);
        den += std::norm(ref[i]);
      }
      CHECK(std::sqrt(num / den) < 2e-6, "thread %d %s transform %zu: rel-L2 %.3g", tid, j.what, b, std::sqrt(num / den));
    }
    // a copy of the plan (own scratch) computes the same bits
    auto copy = plans[k];
    std::complex<float>* out2 = nullptr;
    CHECK(hipMalloc(&out2, count * 8) == hipSuccess, "hipMalloc");
    copy.compute_forward(in, out2).wait();
    std::vector<std::complex<float>> a(j.n), c(j.n);
    (void)hipMemcpy(a.data(), out + (j.batch - 1) * j.n, j.n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c.data(), out2 + (j.batch - 1) * j.n, j.n * 8, hipMemcpyDeviceToHost);
    CHECK(std::memcmp(a.data(), c.data(), j.n * 8) == 0, "thread %d %s: copy of the plan differs", tid, j.what);
    (void)hipFree(in);
    (void)hipFree(out);
    (void)hipFree(out2);
  }
  plans.clear();
  (void)hipStreamDestroy(stream);
}

/// no device: the host-only entry points and the runtime compiler from several threads at once
void host_worker(int tid, barrier_t* bar) {
  bar->arrive_and_wait();
  for (int it = 0; it < 6; ++it) {
    portfft::descriptor<float, portfft::domain::COMPLEX> d({static_cast<std::size_t>(16 + it), 33});
    d.number_of_transforms = 3 + tid;
    CHECK(d.get_input_count(portfft::direction::FORWARD) == (16 + it) * 33 * (3 + tid), "count");
    pfft_desc_t c{};
    const uint64_t len[1] = {static_cast<uint64_t>(3000 + 8 * it)};
    CHECK(pfft_desc_init(&c, PFFT_PRECISION_F32, PFFT_DOMAIN_COMPLEX, 1, len) == PFFT_OK, "init");
    CHECK(pfft_desc_validate(&c) == PFFT_OK, "validate");
    c.number_of_transforms = 0;  // invalid: every thread fails with its own thread-local message
    CHECK(pfft_desc_validate(&c) == PFFT_INVALID_CONFIGURATION, "invalid batch");
    CHECK(std::strlen(pfft_last_error()) > 0, "message");
    // the runtime compiler: the threads ask for the same and for different configurations (process cache, disk cache)
    for (long long n : {3000ll, 3000ll + 8 * (tid % 2), 1200ll}) {
      pfa::wg_params p;
      CHECK(pfa::choose_spec_params(0, n, 160 * 1024, &p), "planned %lld", n);
      std::size_t bytes = 0;
      std::string why;
      CHECK(pfa::jit_compile_only(p, 0, "gfx950", &bytes, &why) && bytes > 0, "compile %lld: %s", n, why.c_str());
    }
  }
}

}  // namespace

int main(int argc, char** argv) {
  std::setvbuf(stdout, nullptr, _IONBF, 0);
  if (argc > 1 && std::string(argv[1]) == "host") {
    const int n = 4;
    barrier_t bar(n);
    std::vector<std::thread> th;
    for (int t = 0; t < n; ++t) th.emplace_back(host_worker, t, &bar);
    for (auto& t : th) t.join();
    long long compiled = 0, from_disk = 0;
    pfa::jit_stats(&compiled, &from_disk);
    std::printf("host threads: %lld kernels compiled, %lld read from the disk cache, %d failures\n", compiled, from_disk,
                g_fail.load());
    if (g_fail == 0) std::printf("multi device host OK\n");
    return g_fail == 0 ? 0 : 1;
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
    std::printf("no device\n");
    return 2;
  }
  const int n_threads = n_dev > 1 ? n_dev : static_cast<int>(env_or("MDT_THREADS", 4));
  std::printf("%d device(s), %d host thread(s)%s\n", n_dev, n_threads, n_dev > 1 ? "" : " (4 streams of device 0)");
  barrier_t bar(n_threads);
  std::vector<double> ms(static_cast<std::size_t>(n_threads) * 3, 0.0);
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t) th.emplace_back(worker, t, n_dev > 1 ? t : 0, n_threads, &bar, &ms);
  for (auto& t : th) t.join();
  for (int t = 0; t < n_threads; ++t) {
    std::printf("thread %d (device %d): N=4096 x %d %.3f ms, N=3000 x 1024 %.3f ms, N=2^18 x %zu %.3f ms per execute\n", t,
                n_dev > 1 ? t : 0, 65536 / n_threads, ms[3 * t], ms[3 * t + 1], env_or("MDT_XCD_BATCH", 256), ms[3 * t + 2]);
  }
  if (g_fail == 0) std::printf("multi device OK\n");
  return g_fail == 0 ? 0 : 1;
}
