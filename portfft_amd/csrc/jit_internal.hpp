// Shared by the two halves of the runtime specialisation (jit_planner.cpp: what to build; jit.cpp: building, caching and
// launching it).  Not part of the library's interface (jit.hpp is).
#pragma once
#include <cstdlib>

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

/// Every environment variable the runtime specialisation looks at, read in ONE place: `jit_knobs::from_env()` at the top
/// of a planner or compiler entry point -- i.e. while a descriptor is being committed, never while one executes (the
/// plan-level knobs are plan_knobs, plan.hpp; INTEGRATION.md lists both).  Strings point into the environment and are
/// used before the entry point returns.
struct jit_knobs {
  bool jit = true;                        ///< PFFT_JIT=0: no runtime specialisation
  bool verbose = false;                   ///< PFFT_JIT_VERBOSE
  const char* cache_dir = nullptr;        ///< PFFT_JIT_CACHE_DIR (may be empty: no disk cache)
  bool no_tuned_table = false;            ///< PFFT_NO_TUNED_TABLE
  bool plan_measure = false;              ///< PFFT_PLAN_MEASURE
  bool fused_nd = true;                   ///< PFFT_FUSED_ND=0
  int max_prime = 61;                     ///< PFFT_JIT_MAX_PRIME (2 ... 61; looked at once per process: it decides which lengths are accepted)
  int stw_mode = 0;                       ///< PFFT_JIT_STW_MODE (1 / 2; 0: by LDS headroom)
  // packed planner
  const char* spec_radices = nullptr;     ///< PFFT_JIT_SPEC_RADICES=n:r0xr1x...
  bool no_prime_lanes = false;            ///< PFFT_NO_PRIME_LANES
  int prime_lanes_min = 37;               ///< PFFT_PRIME_LANES_MIN
  int force_tpf = 0;                      ///< PFFT_JIT_FORCE_TPF
  const char* force_pad = nullptr;        ///< PFFT_JIT_FORCE_PAD
  // register-resident planner
  long hx_min_kib = -1;                   ///< PFFT_JIT_HX_MIN_KIB
  bool hx_pairs = true;                   ///< PFFT_JIT_HX_PAIRS=0
  long hx_pair_min_kib = 80;              ///< PFFT_JIT_HX_PAIR_MIN_KIB
  int hx_pair_gpw = 1;                    ///< PFFT_JIT_HX_PAIR_GPW
  const char* hx_force = nullptr;         ///< PFFT_JIT_HX_FORCE=lanes[:r0xr1x...]
  // strided / rows2d planners
  const char* strided_force = nullptr;    ///< PFFT_JIT_STRIDED_FORCE
  int strided_fpw = 0;                    ///< PFFT_JIT_STRIDED_FPW
  long strided_lds_kib = 0;               ///< PFFT_JIT_STRIDED_LDS_KIB
  int strided_wg = 0;                     ///< PFFT_JIT_STRIDED_WG
  const char* rows2d_force = nullptr;     ///< PFFT_JIT_ROWS2D_FORCE
  bool strided_hx = true;                 ///< PFFT_JIT_STRIDED_HX=0: no register-resident strided kernels
  long strided_hx_min_kib = 80;           ///< PFFT_JIT_STRIDED_HX_MIN_KIB: groups above this take the half-image form
  bool strided_hx_column_rule = true;     ///< PFFT_JIT_STRIDED_HX_COLUMN_RULE=0: the 80 KiB threshold also for column-shaped stages without the modifier
  const char* strided_hx_force = nullptr; ///< PFFT_JIT_STRIDED_HX_FORCE=tpf:per_cu
  bool strided_hx_wide = true;            ///< PFFT_JIT_STRIDED_HX_WIDE=0: no one-per-CU register-resident groups beyond the LDS
  long strided_hx_wide_slack = 8;         ///< PFFT_JIT_STRIDED_HX_WIDE_SLACK: registers beyond the estimate a wide plan may be handed to the compiler with
  long strided_hx_wide_scratch = 192;      ///< PFFT_JIT_STRIDED_HX_WIDE_SCRATCH: bytes of scratch per lane such a kernel may need (see jit.cpp)

  static jit_knobs from_env() {
    jit_knobs k;
    auto str = [](const char* name) { return std::getenv(name); };
    auto num = [&](const char* name, long dflt) {
      const char* e = str(name);
      return e != nullptr ? std::atol(e) : dflt;
    };
    if (const char* e = str("PFFT_JIT")) k.jit = e[0] != '0';
    if (const char* e = str("PFFT_JIT_VERBOSE")) k.verbose = e[0] != '\0' && e[0] != '0';
    k.cache_dir = str("PFFT_JIT_CACHE_DIR");
    k.no_tuned_table = str("PFFT_NO_TUNED_TABLE") != nullptr;
    k.plan_measure = num("PFFT_PLAN_MEASURE", 0) != 0;
    if (const char* e = str("PFFT_FUSED_ND")) k.fused_nd = e[0] != '0';
    k.max_prime = static_cast<int>(num("PFFT_JIT_MAX_PRIME", 61));
    k.max_prime = k.max_prime < 2 ? 2 : (k.max_prime > 61 ? 61 : k.max_prime);
    if (const char* e = str("PFFT_JIT_STW_MODE")) k.stw_mode = std::atoi(e) == 1 ? 1 : 2;
    k.spec_radices = str("PFFT_JIT_SPEC_RADICES");
    k.no_prime_lanes = str("PFFT_NO_PRIME_LANES") != nullptr;
    k.prime_lanes_min = static_cast<int>(num("PFFT_PRIME_LANES_MIN", 37));
    k.force_tpf = static_cast<int>(num("PFFT_JIT_FORCE_TPF", 0));
    k.force_pad = str("PFFT_JIT_FORCE_PAD");
    k.hx_min_kib = num("PFFT_JIT_HX_MIN_KIB", -1);
    if (const char* e = str("PFFT_JIT_HX_PAIRS")) k.hx_pairs = e[0] != '0';
    k.hx_pair_min_kib = num("PFFT_JIT_HX_PAIR_MIN_KIB", 80);
    k.hx_pair_gpw = static_cast<int>(num("PFFT_JIT_HX_PAIR_GPW", 1));
    k.hx_force = str("PFFT_JIT_HX_FORCE");
    k.strided_force = str("PFFT_JIT_STRIDED_FORCE");
    k.strided_fpw = static_cast<int>(num("PFFT_JIT_STRIDED_FPW", 0));
    k.strided_lds_kib = num("PFFT_JIT_STRIDED_LDS_KIB", 0);
    k.strided_wg = static_cast<int>(num("PFFT_JIT_STRIDED_WG", 0));
    k.rows2d_force = str("PFFT_JIT_ROWS2D_FORCE");
    if (const char* e = str("PFFT_JIT_STRIDED_HX")) k.strided_hx = e[0] != '0';
    k.strided_hx_min_kib = num("PFFT_JIT_STRIDED_HX_MIN_KIB", 80);
    if (const char* e = str("PFFT_JIT_STRIDED_HX_COLUMN_RULE")) k.strided_hx_column_rule = e[0] != '0';
    k.strided_hx_force = str("PFFT_JIT_STRIDED_HX_FORCE");
    if (const char* e = str("PFFT_JIT_STRIDED_HX_WIDE")) k.strided_hx_wide = e[0] != '0';
    k.strided_hx_wide_scratch = num("PFFT_JIT_STRIDED_HX_WIDE_SCRATCH", 192);
    k.strided_hx_wide_slack = num("PFFT_JIT_STRIDED_HX_WIDE_SLACK", 8);
    return k;
  }
};

}  // namespace pfa
