// portfft/portfft.hpp -- header-only C++17 facade over the C ABI of portfft_amd.h.
//
// Gives C++ callers the reference's plan-and-commit interface unchanged in spirit:
//   portfft::descriptor<Scalar, Domain>           (/root/reference/src/portfft/descriptor.hpp:43-271)
//   portfft::committed_descriptor<Scalar, Domain> (/root/reference/src/portfft/committed_descriptor.hpp:46-315)
//   enums                                          (/root/reference/src/portfft/enums.hpp:25-38)
//   exceptions                                     (/root/reference/src/portfft/common/exceptions.hpp:32-77)
// with these substitutions: sycl::queue -> portfft::queue (a HIP stream), sycl::event -> portfft::event (stream-
// ordered completion; wait() blocks), USM pointers -> HIP device pointers.  sycl::buffer overloads do not exist
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

/// Stands where the reference takes a sycl::queue: an in-order HIP stream (nullptr = the default stream).
class queue {
 public:
  queue() = default;
  explicit queue(void* hip_stream) : stream_(hip_stream) {}
  void* native() const { return stream_; }

 private:
  void* stream_ = nullptr;
};

/// Stands where the reference returns a sycl::event.  Work is ordered on the plan's stream; wait() blocks until
/// everything enqueued so far on that stream has finished.
class event {
 public:
  event() = default;
  explicit event(std::shared_ptr<pfft_plan_t> plan) : plan_(std::move(plan)) {}
  void wait() const {
    if (plan_) detail::check(pfft_plan_wait(plan_.get()));
  }

 private:
  std::shared_ptr<pfft_plan_t> plan_;
};

template <typename Scalar, domain Domain>
struct descriptor;

template <typename Scalar, domain Domain>
class committed_descriptor {
  static_assert(std::is_same_v<Scalar, float> || std::is_same_v<Scalar, double>, "Scalar must be float or double");
  friend struct descriptor<Scalar, Domain>;
  std::shared_ptr<pfft_plan_t> plan_;

  committed_descriptor(const pfft_desc_t& d, queue& q) {
    pfft_plan_t* p = nullptr;
    detail::check(pfft_plan_create(&d, q.native(), &p));
    plan_ = std::shared_ptr<pfft_plan_t>(p, [](pfft_plan_t* x) { (void)pfft_plan_destroy(x); });
  }

  event run(direction dir, const void* in, void* out) {
    detail::check(pfft_execute(plan_.get(), static_cast<int32_t>(dir), in, out));
    return event(plan_);
  }
  event run_split(direction dir, const void* ir, const void* ii, void* outr, void* outi) {
    detail::check(pfft_execute_split(plan_.get(), static_cast<int32_t>(dir), ir, ii, outr, outi));
    return event(plan_);
  }

 public:
  using complex_type = std::complex<Scalar>;
  using scalar_type = Scalar;

  // dependencies are expressed by stream order; the vector overloads of the reference take explicit events
  // (committed_descriptor.hpp:171-310), here callers enqueue on the same stream or wait() first.

  /// in-place, interleaved (committed_descriptor.hpp:171-176 / 215-218)
  event compute_forward(complex_type* inout) { return run(direction::FORWARD, inout, inout); }
  event compute_backward(complex_type* inout) { return run(direction::BACKWARD, inout, inout); }
  /// in-place, split (committed_descriptor.hpp:186-192 / 228-232)
  event compute_forward(scalar_type* inout_real, scalar_type* inout_imag) {
    return run_split(direction::FORWARD, inout_real, inout_imag, inout_real, inout_imag);
  }
  event compute_backward(scalar_type* inout_real, scalar_type* inout_imag) {
    return run_split(direction::BACKWARD, inout_real, inout_imag, inout_real, inout_imag);
  }
  /// out-of-place, interleaved (committed_descriptor.hpp:242-246 / 288-293)
  event compute_forward(const complex_type* in, complex_type* out) { return run(direction::FORWARD, in, out); }
  event compute_backward(const complex_type* in, complex_type* out) { return run(direction::BACKWARD, in, out); }
  /// out-of-place, split (committed_descriptor.hpp:258-263 / 305-310)
  event compute_forward(const scalar_type* in_real, const scalar_type* in_imag, scalar_type* out_real,
                        scalar_type* out_imag) {
    return run_split(direction::FORWARD, in_real, in_imag, out_real, out_imag);
  }
  event compute_backward(const scalar_type* in_real, const scalar_type* in_imag, scalar_type* out_real,
                         scalar_type* out_imag) {
    return run_split(direction::BACKWARD, in_real, in_imag, out_real, out_imag);
  }
  /// real-to-complex entry points exist in the reference only to throw (committed_descriptor.hpp:134-137,273-278)
  event compute_forward(const scalar_type*, complex_type*) {
    throw unsupported_configuration("Real to complex FFTs not yet implemented.");
  }
  event compute_backward(const complex_type*, scalar_type*) {
    throw unsupported_configuration("Complex to real FFTs not yet implemented.");
  }

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
