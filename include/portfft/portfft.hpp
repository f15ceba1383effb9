// portfft/portfft.hpp -- header-only C++17 facade over the C ABI of portfft_amd.h.
//
// Gives C++ callers the reference's plan-and-commit interface unchanged in spirit:
//   portfft::descriptor<Scalar, Domain>           (/root/reference/src/portfft/descriptor.hpp:43-271)
//   portfft::committed_descriptor<Scalar, Domain> (/root/reference/src/portfft/committed_descriptor.hpp:46-315)
//   enums                                          (/root/reference/src/portfft/enums.hpp:25-38)
//   exceptions                                     (/root/reference/src/portfft/common/exceptions.hpp:32-77)
// with these substitutions: sycl::queue -> portfft::queue (a HIP stream), sycl::event -> portfft::event (a hipEvent_t
// recorded behind the submission: per-call completion, usable as a dependency of later calls), USM pointers -> HIP
// device pointers.  sycl::buffer overloads do not exist
// (HIP has no buffer/accessor model).  Link with -lportfft_amd.
#ifndef PORTFFT_PORTFFT_HPP
#define PORTFFT_PORTFFT_HPP

#include <complex>
#include <cstddef>
#include <functional>
#include <memory>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../portfft_amd.h"

namespace portfft {

enum class domain { REAL = PFFT_DOMAIN_REAL, COMPLEX = PFFT_DOMAIN_COMPLEX };
enum class complex_storage { INTERLEAVED_COMPLEX = PFFT_INTERLEAVED_COMPLEX, SPLIT_COMPLEX = PFFT_SPLIT_COMPLEX };
enum class placement { IN_PLACE = PFFT_IN_PLACE, OUT_OF_PLACE = PFFT_OUT_OF_PLACE };
enum class direction { FORWARD = PFFT_FORWARD, BACKWARD = PFFT_BACKWARD };
constexpr direction inv(direction dir) { return dir == direction::FORWARD ? direction::BACKWARD : direction::FORWARD; }

class base_error : public std::runtime_error {
 public:
  explicit base_error(const std::string& what) : std::runtime_error(what) {}
};
struct internal_error : public base_error {
  using base_error::base_error;
};
struct invalid_configuration : public base_error {
  using base_error::base_error;
};
struct unsupported_configuration : public base_error {
  using base_error::base_error;
};
struct out_of_local_memory_error : public unsupported_configuration {
  using unsupported_configuration::unsupported_configuration;
};
/// a HIP runtime call failed inside the library (the reference would surface a sycl::exception)
struct device_error : public base_error {
  using base_error::base_error;
};

namespace detail {
inline void check(pfft_status st) {
  if (st == PFFT_OK) return;
  const std::string msg = pfft_last_error();
  switch (st) {
    case PFFT_INVALID_CONFIGURATION:
      throw invalid_configuration(msg);
    case PFFT_UNSUPPORTED_CONFIGURATION:
      throw unsupported_configuration(msg);
    case PFFT_OUT_OF_LOCAL_MEMORY:
      throw out_of_local_memory_error(msg);
    case PFFT_HIP_ERROR:
      throw device_error(msg);
    default:
      throw internal_error(msg);
  }
}

inline std::vector<std::size_t> get_default_strides(const std::vector<std::size_t>& lengths) {
  std::vector<std::size_t> strides(lengths.size());
  std::size_t total = 1;
  for (std::size_t i = lengths.size(); i-- > 0;) {
    strides[i] = total;
    total *= lengths[i];
  }
  return strides;
}
}  // namespace detail

template <typename Scalar, domain Domain>
class committed_descriptor;
class queue;

/// Stands where the reference returns a sycl::event: the completion of ONE submission.  Owns a hipEvent_t recorded
/// on the plan's stream behind the last kernel of the compute_* call that returned it; copies share it.  wait()
/// blocks the host until that submission has finished (hipEventSynchronize); passing the event in another call's
/// `dependencies` orders that call behind it on the device (hipStreamWaitEvent), also across streams.  A
/// default-constructed event is already complete.  native() is the hipEvent_t for direct HIP interop.
class event {
 public:
  event() = default;
  void wait() const {
    if (ev_) detail::check(pfft_event_wait(ev_.get()));
  }
  /// sycl::event::wait_and_throw(): errors surface as exceptions from wait() already
  void wait_and_throw() const { wait(); }
  /// true once the submission has finished (info::event_command_status::complete in the reference's world)
  bool is_complete() const {
    int32_t done = 1;
    if (ev_) detail::check(pfft_event_query(ev_.get(), &done));
    return done != 0;
  }
  void* native() const { return ev_.get(); }
  /// sycl::event::wait(const std::vector<event>&)
  static void wait(const std::vector<event>& events) {
    for (const event& e : events) e.wait();
  }

 private:
  template <typename S, domain D>
  friend class committed_descriptor;
  friend class queue;
  explicit event(void* hip_event) : ev_(hip_event, [](void* e) { (void)pfft_event_destroy(e); }) {}
  std::shared_ptr<void> ev_;
};

/// Stands where the reference takes a sycl::queue: an in-order HIP stream (nullptr = the default stream).  copy()
/// and wait() are the two queue members the reference's own callers use around compute_* (test/unit_test/
/// fft_test_utils.hpp:286-333, test/bench/portfft/launch_bench.hpp:96-135).
class queue {
 public:
  queue() = default;
  explicit queue(void* hip_stream) : stream_(hip_stream) {}
  void* native() const { return stream_; }

  /// sycl::queue::copy(src, dest, count, dependencies): asynchronous on this stream, any combination of host and
  /// device pointers; the returned event completes with the copy
  template <typename T>
  event copy(const T* src, T* dest, std::size_t count, const std::vector<event>& dependencies = {}) {
    std::vector<void*> deps;
    deps.reserve(dependencies.size());
    for (const event& e : dependencies) deps.push_back(e.native());
    void* ev = nullptr;
    detail::check(pfft_queue_copy(stream_, src, dest, count * sizeof(T), static_cast<int32_t>(deps.size()), deps.data(),
                                  &ev));
    return event(ev);
  }
  /// sycl::queue::wait() / wait_and_throw()
  void wait() const { detail::check(pfft_queue_wait(stream_)); }
  void wait_and_throw() const { wait(); }

 private:
  void* stream_ = nullptr;
};

template <typename Scalar, domain Domain>
struct descriptor;

template <typename Scalar, domain Domain>
class committed_descriptor {
  static_assert(std::is_same_v<Scalar, float> || std::is_same_v<Scalar, double>, "Scalar must be float or double");
  friend struct descriptor<Scalar, Domain>;
  std::shared_ptr<pfft_plan_t> plan_;

  static std::shared_ptr<pfft_plan_t> own(pfft_plan_t* p) {
    return std::shared_ptr<pfft_plan_t>(p, [](pfft_plan_t* x) { (void)pfft_plan_destroy(x); });
  }

  committed_descriptor(const pfft_desc_t& d, queue& q) {
    pfft_plan_t* p = nullptr;
    detail::check(pfft_plan_create(&d, q.native(), &p));
    plan_ = own(p);
  }

  static std::vector<void*> natives(const std::vector<event>& dependencies) {
    std::vector<void*> deps;
    deps.reserve(dependencies.size());
    for (const event& e : dependencies) deps.push_back(e.native());
    return deps;
  }

  event run(direction dir, const void* in, void* out, const std::vector<event>& dependencies) {
    const std::vector<void*> deps = natives(dependencies);
    void* ev = nullptr;
    detail::check(pfft_execute_ex(plan_.get(), static_cast<int32_t>(dir), in, out, static_cast<int32_t>(deps.size()),
                                  deps.data(), &ev));
    return event(ev);
  }
  event run_split(direction dir, const void* ir, const void* ii, void* outr, void* outi,
                  const std::vector<event>& dependencies) {
    const std::vector<void*> deps = natives(dependencies);
    void* ev = nullptr;
    detail::check(pfft_execute_split_ex(plan_.get(), static_cast<int32_t>(dir), ir, ii, outr, outi,
                                        static_cast<int32_t>(deps.size()), deps.data(), &ev));
    return event(ev);
  }

 public:
  using complex_type = std::complex<Scalar>;
  using scalar_type = Scalar;

  /// Copies share the kernels and twiddle tables and get scratch memory of their own, like the reference's
  /// (committed_descriptor_impl.hpp:774-817): two copies can execute concurrently on two host threads / streams.
  committed_descriptor(const committed_descriptor& other) {
    pfft_plan_t* p = nullptr;
    detail::check(pfft_plan_clone(other.plan_.get(), &p));
    plan_ = own(p);
  }
  committed_descriptor& operator=(const committed_descriptor& other) {
    if (this != &other) {
      pfft_plan_t* p = nullptr;
      detail::check(pfft_plan_clone(other.plan_.get(), &p));
      plan_ = own(p);
    }
    return *this;
  }
  committed_descriptor(committed_descriptor&&) noexcept = default;
  committed_descriptor& operator=(committed_descriptor&&) noexcept = default;

  // Signatures of the USM overloads follow committed_descriptor.hpp:171-310 argument for argument:
  // `dependencies` are events that must complete before the computation starts; the returned event completes with
  // this computation.

  /// in-place, interleaved (committed_descriptor.hpp:171-176 / 215-218)
  event compute_forward(complex_type* inout, const std::vector<event>& dependencies = {}) {
    return run(direction::FORWARD, inout, inout, dependencies);
  }
  event compute_backward(complex_type* inout, const std::vector<event>& dependencies = {}) {
    return run(direction::BACKWARD, inout, inout, dependencies);
  }
  /// in-place, split (committed_descriptor.hpp:186-192 / 228-232)
  event compute_forward(scalar_type* inout_real, scalar_type* inout_imag,
                        const std::vector<event>& dependencies = {}) {
    return run_split(direction::FORWARD, inout_real, inout_imag, inout_real, inout_imag, dependencies);
  }
  event compute_backward(scalar_type* inout_real, scalar_type* inout_imag,
                         const std::vector<event>& dependencies = {}) {
    return run_split(direction::BACKWARD, inout_real, inout_imag, inout_real, inout_imag, dependencies);
  }
  /// out-of-place, interleaved (committed_descriptor.hpp:242-246 / 288-293)
  event compute_forward(const complex_type* in, complex_type* out, const std::vector<event>& dependencies = {}) {
    return run(direction::FORWARD, in, out, dependencies);
  }
  event compute_backward(const complex_type* in, complex_type* out, const std::vector<event>& dependencies = {}) {
    return run(direction::BACKWARD, in, out, dependencies);
  }
  /// out-of-place, split (committed_descriptor.hpp:258-263 / 305-310)
  event compute_forward(const scalar_type* in_real, const scalar_type* in_imag, scalar_type* out_real,
                        scalar_type* out_imag, const std::vector<event>& dependencies = {}) {
    return run_split(direction::FORWARD, in_real, in_imag, out_real, out_imag, dependencies);
  }
  event compute_backward(const scalar_type* in_real, const scalar_type* in_imag, scalar_type* out_real,
                         scalar_type* out_imag, const std::vector<event>& dependencies = {}) {
    return run_split(direction::BACKWARD, in_real, in_imag, out_real, out_imag, dependencies);
  }
  /// real-to-complex entry points exist in the reference only to throw (committed_descriptor.hpp:134-137,273-278)
  event compute_forward(const scalar_type*, complex_type*, const std::vector<event>& = {}) {
    throw unsupported_configuration("Real to complex FFTs not yet implemented.");
  }
  event compute_backward(const complex_type*, scalar_type*, const std::vector<event>& = {}) {
    throw unsupported_configuration("Complex to real FFTs not yet implemented.");
  }

  /// queue.wait() of the reference's callers: everything submitted on the plan's stream has finished
  void wait() const { detail::check(pfft_plan_wait(plan_.get())); }


  pfft_plan_info_t info() const {
    pfft_plan_info_t i{};
    detail::check(pfft_plan_get_info(plan_.get(), &i));
    return i;
  }
};

template <typename DescScalar, domain DescDomain>
struct descriptor {
  using Scalar = DescScalar;
  static_assert(std::is_floating_point_v<Scalar>, "Scalar must be a floating point type");
  static constexpr domain Domain = DescDomain;

  std::vector<std::size_t> lengths;
  Scalar forward_scale = 1;
  Scalar backward_scale = 1;
  std::size_t number_of_transforms = 1;
  portfft::complex_storage complex_storage = portfft::complex_storage::INTERLEAVED_COMPLEX;
  portfft::placement placement = portfft::placement::OUT_OF_PLACE;
  std::vector<std::size_t> forward_strides;
  std::vector<std::size_t> backward_strides;
  std::size_t forward_distance = 1;
  std::size_t backward_distance = 1;
  std::size_t forward_offset = 0;
  std::size_t backward_offset = 0;

  explicit descriptor(const std::vector<std::size_t>& lengths)
      : lengths(lengths), forward_strides(detail::get_default_strides(lengths)), backward_strides(forward_strides) {
    const std::size_t total = get_flattened_length();
    forward_distance = total;
    backward_distance = total;
  }

  /// validate, then plan (descriptor.hpp:152-156)
  committed_descriptor<Scalar, Domain> commit(queue& q) {
    const pfft_desc_t d = to_c();
    detail::check(pfft_desc_validate(&d));
    return committed_descriptor<Scalar, Domain>(d, q);
  }

  std::size_t get_flattened_length() const noexcept {
    return std::accumulate(lengths.begin(), lengths.end(), std::size_t{1}, std::multiplies<std::size_t>());
  }
  std::size_t get_input_count(direction dir) const {
    const pfft_desc_t d = to_c();
    return static_cast<std::size_t>(pfft_desc_input_count(&d, static_cast<int32_t>(dir)));
  }
  std::size_t get_output_count(direction dir) const { return get_input_count(inv(dir)); }

  const std::vector<std::size_t>& get_strides(direction dir) const noexcept {
    return dir == direction::FORWARD ? forward_strides : backward_strides;
  }
  std::vector<std::size_t>& get_strides(direction dir) noexcept {
    return dir == direction::FORWARD ? forward_strides : backward_strides;
  }
  std::size_t get_distance(direction dir) const noexcept {
    return dir == direction::FORWARD ? forward_distance : backward_distance;
  }
  std::size_t& get_distance(direction dir) noexcept {
    return dir == direction::FORWARD ? forward_distance : backward_distance;
  }
  std::size_t get_offset(direction dir) const noexcept {
    return dir == direction::FORWARD ? forward_offset : backward_offset;
  }
  std::size_t& get_offset(direction dir) noexcept {
    return dir == direction::FORWARD ? forward_offset : backward_offset;
  }
  Scalar get_scale(direction dir) const noexcept { return dir == direction::FORWARD ? forward_scale : backward_scale; }
  Scalar& get_scale(direction dir) noexcept { return dir == direction::FORWARD ? forward_scale : backward_scale; }

 private:
  pfft_desc_t to_c() const {
    if (lengths.size() > PFFT_MAX_RANK) {
      throw unsupported_configuration("At most " + std::to_string(PFFT_MAX_RANK) + " dimensions are supported");
    }
    pfft_desc_t d{};
    d.precision = std::is_same_v<Scalar, double> ? PFFT_PRECISION_F64 : PFFT_PRECISION_F32;
    d.domain = static_cast<int32_t>(Domain);
    d.rank = static_cast<int32_t>(lengths.size());
    d.complex_storage = static_cast<int32_t>(complex_storage);
    d.placement = static_cast<int32_t>(placement);
    d.n_forward_strides = static_cast<int32_t>(forward_strides.size());
    d.n_backward_strides = static_cast<int32_t>(backward_strides.size());
    for (std::size_t i = 0; i < lengths.size(); ++i) d.lengths[i] = lengths[i];
    for (std::size_t i = 0; i < forward_strides.size() && i < PFFT_MAX_RANK; ++i) d.forward_strides[i] = forward_strides[i];
    for (std::size_t i = 0; i < backward_strides.size() && i < PFFT_MAX_RANK; ++i) d.backward_strides[i] = backward_strides[i];
    d.forward_distance = forward_distance;
    d.backward_distance = backward_distance;
    d.forward_offset = forward_offset;
    d.backward_offset = backward_offset;
    d.number_of_transforms = number_of_transforms;
    d.forward_scale = static_cast<double>(forward_scale);
    d.backward_scale = static_cast<double>(backward_scale);
    return d;
  }
};

}  // namespace portfft

#endif  // PORTFFT_PORTFFT_HPP
