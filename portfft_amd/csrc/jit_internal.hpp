// Shared by the two halves of the runtime specialisation (jit_planner.cpp: what to build; jit.cpp: building, caching and
// launching it).  Not part of the library's interface (jit.hpp is).
#pragma once
#include "../../include/portfft_amd.h"

namespace pfa {

/// most passes a runtime-specialised work-group kernel may have (wg_cfg radix lists)
constexpr int MAX_PASSES = 6;

inline int elem_bytes_of(int precision) { return precision == PFFT_PRECISION_F64 ? 16 : 8; }

inline bool is_prime_i(int v) {
  if (v < 2) return false;
  for (int i = 2; i * i <= v; ++i) {
    if (v % i == 0) return false;
  }
  return true;
}

}  // namespace pfa
