// Host-side descriptor helpers (see descriptor.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "common.hpp"

namespace pfa {

/// strides / distance / offset of one domain of a descriptor
struct view_t {
  std::vector<uint64_t> strides;
  int n_strides = 0;
  uint64_t distance = 0;
  uint64_t offset = 0;
};

std::vector<uint64_t> default_strides(const pfft_desc_t& d);
uint64_t flattened_length(const pfft_desc_t& d);
view_t view_of(const pfft_desc_t& d, int direction);
/// elements a buffer of domain `direction` must hold (descriptor::get_input_count)
uint64_t buffer_count(const pfft_desc_t& d, int direction);
int layout_of(const pfft_desc_t& d, int direction);
/// throws pfa::error(invalid / unsupported) like detail::validate::validate_descriptor
void validate(const pfft_desc_t& d);
int64_t largest_factor_le(int64_t n, int64_t limit);
bool fits_wavefront_registers(int64_t n, int scalar_bytes);
const std::string& last_error();

}  // namespace pfa
