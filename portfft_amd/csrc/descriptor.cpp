// Host-side descriptor logic behind the pfft_desc_* entry points: defaults, buffer-size arithmetic, layout
// classification and validation.  No device is needed for anything in this file.
//
// Mirrors (behaviour, messages and exception kinds):
//   /root/reference/src/portfft/descriptor.hpp:131-183,262-270   (defaults, get_input_count / get_output_count)
//   /root/reference/src/portfft/utils.hpp:190-246                (default strides, layout classification)
//   /root/reference/src/portfft/descriptor_validation.hpp:38-281 (validation)
#include "descriptor.hpp"

#include <algorithm>
#include <numeric>
#include <vector>

namespace pfa {

namespace {
thread_local std::string g_last_error;
}
void set_last_error(const std::string& msg) { g_last_error = msg; }
const std::string& last_error() { return g_last_error; }

std::vector<uint64_t> default_strides(const pfft_desc_t& d) {
  std::vector<uint64_t> s(static_cast<size_t>(std::max(d.rank, 0)));
  uint64_t total = 1;
  for (int i = d.rank - 1; i >= 0; --i) {
    s[static_cast<size_t>(i)] = total;
    total *= d.lengths[i];
  }
  return s;
}

uint64_t flattened_length(const pfft_desc_t& d) {
  uint64_t t = 1;
  for (int i = 0; i < d.rank; ++i) t *= d.lengths[i];
  return t;
}

view_t view_of(const pfft_desc_t& d, int direction) {
  view_t v;
  const bool fwd = direction == PFFT_FORWARD;
  const int n = fwd ? d.n_forward_strides : d.n_backward_strides;
  const uint64_t* s = fwd ? d.forward_strides : d.backward_strides;
  v.strides.assign(s, s + std::min(std::max(n, 0), PFFT_MAX_RANK));
  v.n_strides = n;
  v.distance = fwd ? d.forward_distance : d.backward_distance;
  v.offset = fwd ? d.forward_offset : d.backward_offset;
  return v;
}

uint64_t buffer_count(const pfft_desc_t& d, int direction) {
  const view_t v = view_of(d, direction);
  uint64_t last = (d.number_of_transforms - 1) * v.distance;
  for (int i = 0; i < d.rank && i < static_cast<int>(v.strides.size()); ++i) {
    last += (d.lengths[i] - 1) * v.strides[static_cast<size_t>(i)];
  }
  return v.offset + last + 1;
}

int layout_of(const pfft_desc_t& d, int direction) {
  const view_t v = view_of(d, direction);
  if (v.n_strides == d.rank && v.strides == default_strides(d) && v.distance == flattened_length(d)) {
    return PFFT_LAYOUT_PACKED;
  }
  if (d.rank == 1 && v.distance == 1 && !v.strides.empty() && v.strides.back() == d.number_of_transforms) {
    return PFFT_LAYOUT_BATCH_INTERLEAVED;
  }
  return PFFT_LAYOUT_UNPACKED;
}

namespace {

void check_domain_view(const pfft_desc_t& d, int direction, const char* name) {
  const view_t v = view_of(d, direction);
  if (v.n_strides != d.rank) {
    fail(PFFT_INVALID_CONFIGURATION, "Mismatching ", name, " strides length got ", v.n_strides, " expected ", d.rank);
  }
  for (int i = 0; i < d.rank; ++i) {
    if (v.strides[static_cast<size_t>(i)] == 0) {
      fail(PFFT_INVALID_CONFIGURATION, "Invalid ", name, " stride[", i, "]=0, must be positive");
    }
  }
  if (d.number_of_transforms > 1 && v.distance == 0) {
    fail(PFFT_INVALID_CONFIGURATION, "Invalid ", name, " distance 0, must be positive for batched FFTs");
  }
  if (d.rank > 1) {
    // batches are one more tensor dimension with stride `distance`: sorted by stride, every dimension must fit
    // inside the next one
    std::vector<uint64_t> gs = v.strides;
    std::vector<uint64_t> gn(d.lengths, d.lengths + d.rank);
    if (d.number_of_transforms > 1) {
      gs.push_back(v.distance);
      gn.push_back(d.number_of_transforms);
    }
    std::vector<size_t> order(gs.size());
    std::iota(order.begin(), order.end(), size_t{0});
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return gs[a] < gs[b]; });
    for (size_t i = 1; i < order.size(); ++i) {
      if (gs[order[i - 1]] * gn[order[i - 1]] > gs[order[i]]) {
        fail(PFFT_INVALID_CONFIGURATION, "Domain ", name,
             ": multi-dimension strides are not large enough to avoid overlap");
      }
    }
    return;
  }
  // 1-D: walk the batches whose first element falls on the first batch's stride lattice
  const uint64_t n = d.lengths[0];
  const uint64_t stride = v.strides[0];
  const uint64_t distance = v.distance;
  const uint64_t first_batch_limit = stride * n;
  const uint64_t all_batches_limit = distance * d.number_of_transforms;
  if ((stride <= distance && first_batch_limit <= distance) || (distance <= stride && all_batches_limit <= stride)) {
    return;
  }
  for (uint64_t b = 1; b < d.number_of_transforms;) {
    const uint64_t first_idx = b * distance;
    const uint64_t column = first_idx % stride;
    if (column == 0) {
      if (first_idx >= first_batch_limit) return;
      fail(PFFT_INVALID_CONFIGURATION, "Domain ", name, ": batch ", b, " collides with first batch at index ", first_idx);
    }
    b += (stride - column + distance - 1) / distance;
  }
}

}  // namespace

int64_t largest_factor_le(int64_t n, int64_t limit) {
  for (int64_t i = std::min(limit, n); i > 1; --i) {
    if (n % i == 0) return i;
  }
  return 1;
}

bool fits_wavefront_registers(int64_t n, int scalar_bytes) {
  // reference rule (common/subgroup.hpp:248-253 with common/workitem.hpp:154-185): N = lanes * per_lane with
  // lanes <= sub-group size must leave per_lane (+ temporaries) within 512 bytes of registers.  Evaluated for the
  // reference's default sub-group size 32 so the same descriptors are accepted / rejected.
  const int64_t per_lane = n / largest_factor_le(n, 32);
  // temporaries of the reference's recursive register FFT: N + max over the two factors, recursively
  struct rec {
    static int64_t temps(int64_t m, int level) {
      int64_t f0 = 1;
      for (int64_t i = 2; i * i <= m; ++i) {
        if (m % i == 0) f0 = i;
      }
      const int64_t f1 = m / f0;
      if (f0 < 2 || f1 < 2) return m;
      int64_t a = 2, b = 2;
      if (level < 4) {
        a = temps(f0, level + 1);
        b = temps(f1, level + 1);
      }
      return std::max(a, b) + m;
    }
  };
  return (per_lane + rec::temps(per_lane, 0)) * 2 * scalar_bytes <= 512;
}

void validate(const pfft_desc_t& d) {
  if (d.domain == PFFT_DOMAIN_REAL) fail(PFFT_UNSUPPORTED_CONFIGURATION, "REAL domain is unsupported");
  if (d.domain != PFFT_DOMAIN_COMPLEX) fail(PFFT_INVALID_CONFIGURATION, "Invalid domain ", d.domain);
  if (d.precision != PFFT_PRECISION_F32 && d.precision != PFFT_PRECISION_F64) {
    fail(PFFT_INVALID_CONFIGURATION, "Invalid precision ", d.precision);
  }
  if (d.number_of_transforms == 0) {
    fail(PFFT_INVALID_CONFIGURATION, "Invalid number of transform 0, must be positive");
  }
  if (d.rank <= 0) fail(PFFT_INVALID_CONFIGURATION, "Invalid lengths, must have at least 1 dimension");
  if (d.rank > PFFT_MAX_RANK) {
    fail(PFFT_UNSUPPORTED_CONFIGURATION, "At most ", PFFT_MAX_RANK, " dimensions are supported, got ", d.rank);
  }
  for (int i = 0; i < d.rank; ++i) {
    if (d.lengths[i] == 0) fail(PFFT_INVALID_CONFIGURATION, "Invalid lengths[", i, "]=0, must be positive");
  }
  if (d.placement == PFFT_IN_PLACE) {
    const view_t f = view_of(d, PFFT_FORWARD), b = view_of(d, PFFT_BACKWARD);
    if (f.n_strides != b.n_strides || f.strides != b.strides) {
      fail(PFFT_INVALID_CONFIGURATION, "Invalid forward and backward strides must match for in-place configurations");
    }
    if (f.distance != b.distance) {
      fail(PFFT_INVALID_CONFIGURATION,
           "Invalid forward and backward distances must match for in-place configurations");
    }
    check_domain_view(d, PFFT_FORWARD, "forward");
  } else {
    check_domain_view(d, PFFT_FORWARD, "forward");
    check_domain_view(d, PFFT_BACKWARD, "backward");
  }
  const int fl = layout_of(d, PFFT_FORWARD), bl = layout_of(d, PFFT_BACKWARD);
  if (d.rank > 1 && !(fl == PFFT_LAYOUT_PACKED && bl == PFFT_LAYOUT_PACKED)) {
    fail(PFFT_UNSUPPORTED_CONFIGURATION, "Multi-dimensional transforms are only supported with default data layout");
  }
  // The reference rejects UNPACKED layouts for lengths beyond its subgroup tier ("Arbitrary strides and distances are
  // only supported for sizes that fit in the registers of a subgroup", committed_descriptor_impl.hpp:757-764).  That
  // is a limit of its kernels, not of the interface: here every length a single work-group can hold takes any
  // stride / distance (stockham_wg_unpacked_kernel, generic tier), so the check is not mirrored; only lengths that
  // need the multi-kernel tier still require the default layout (plan_*.cpp).
}

}  // namespace pfa

// ---------------------------------------------------------------------------------------------------------------
// C ABI (descriptor part)
// ---------------------------------------------------------------------------------------------------------------
extern "C" {

pfft_status pfft_desc_init(pfft_desc_t* desc, int32_t precision, int32_t domain, int32_t rank,
                           const uint64_t* lengths) {
  return pfa::guarded([&] {
    if (desc == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null descriptor");
    if (rank < 0 || rank > PFFT_MAX_RANK) {
      pfa::fail(PFFT_UNSUPPORTED_CONFIGURATION, "At most ", PFFT_MAX_RANK, " dimensions are supported, got ", rank);
    }
    if (rank > 0 && lengths == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null lengths");
    *desc = pfft_desc_t{};
    desc->precision = precision;
    desc->domain = domain;
    desc->rank = rank;
    desc->complex_storage = PFFT_INTERLEAVED_COMPLEX;
    desc->placement = PFFT_OUT_OF_PLACE;
    for (int i = 0; i < rank; ++i) desc->lengths[i] = lengths[i];
    const auto s = pfa::default_strides(*desc);
    for (int i = 0; i < rank; ++i) {
      desc->forward_strides[i] = s[static_cast<size_t>(i)];
      desc->backward_strides[i] = s[static_cast<size_t>(i)];
    }
    desc->n_forward_strides = rank;
    desc->n_backward_strides = rank;
    const uint64_t total = pfa::flattened_length(*desc);
    desc->forward_distance = total;
    desc->backward_distance = total;
    desc->forward_offset = 0;
    desc->backward_offset = 0;
    desc->number_of_transforms = 1;
    desc->forward_scale = 1.0;
    desc->backward_scale = 1.0;
  });
}

pfft_status pfft_desc_validate(const pfft_desc_t* desc) {
  return pfa::guarded([&] {
    if (desc == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null descriptor");
    pfa::validate(*desc);
  });
}

uint64_t pfft_desc_flattened_length(const pfft_desc_t* desc) { return pfa::flattened_length(*desc); }

uint64_t pfft_desc_input_count(const pfft_desc_t* desc, int32_t direction) {
  return pfa::buffer_count(*desc, direction);
}

uint64_t pfft_desc_output_count(const pfft_desc_t* desc, int32_t direction) {
  return pfa::buffer_count(*desc, direction == PFFT_FORWARD ? PFFT_BACKWARD : PFFT_FORWARD);
}

int32_t pfft_desc_layout(const pfft_desc_t* desc, int32_t direction) { return pfa::layout_of(*desc, direction); }

const char* pfft_last_error(void) { return pfa::last_error().c_str(); }

const char* pfft_status_string(pfft_status s) {
  switch (s) {
    case PFFT_OK:
      return "ok";
    case PFFT_INVALID_CONFIGURATION:
      return "invalid_configuration";
    case PFFT_UNSUPPORTED_CONFIGURATION:
      return "unsupported_configuration";
    case PFFT_OUT_OF_LOCAL_MEMORY:
      return "out_of_local_memory_error";
    case PFFT_INTERNAL_ERROR:
      return "internal_error";
    case PFFT_HIP_ERROR:
      return "hip_error";
  }
  return "unknown";
}

const char* pfft_version(void) { return "portfft_amd 0.1 (gfx950)"; }

}  // extern "C"
