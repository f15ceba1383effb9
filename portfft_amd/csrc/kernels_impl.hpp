// Shared body of kernels_f32.hip / kernels_f64.hip: turns wg_cfg<...> variants into registry entries.
#pragma once
#include <cstdlib>
#include <vector>

#include <hip/hip_ext.h>

#include "../../include/portfft_amd.h"
#include "generic_kernel.hpp"
#include "kernels.hpp"
#include "stockham_rows2d.hpp"
#include "stockham_strided.hpp"
#include "stockham_wg.hpp"
#include "stockham_wg_hx.hpp"
#include "stockham_xlane.hpp"

#include <tuple>
#include <utility>

namespace pfa {

/// Every launch of a pre-compiled kernel: the ordinary launch, or -- when the submission's completion event is armed
/// (kernels.hpp: arm_stop_event) -- hipExtLaunchKernel with that event as the dispatch's stop event.  The arguments
/// are converted to the kernel's formal parameter types before they are packed.
template <typename... KArgs, typename... Args, size_t... I>
inline hipError_t launch_with_stop_event(void (*kernel)(KArgs...), dim3 g, dim3 b, size_t lds, hipStream_t stream,
                                         hipEvent_t stop, std::index_sequence<I...>, Args&&... args) {
  std::tuple<KArgs...> formal(static_cast<KArgs>(args)...);
  void* p[] = {static_cast<void*>(&std::get<I>(formal))...};
  return hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), g, b, p, lds, stream, nullptr, stop, 0);
}
template <typename... KArgs, typename... Args>
inline void launch_kernel(void (*kernel)(KArgs...), dim3 g, dim3 b, size_t lds, hipStream_t stream, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
  if (hipEvent_t stop = take_stop_event()) {
    (void)launch_with_stop_event(kernel, g, b, lds, stream, stop, std::index_sequence_for<KArgs...>{},
                                 std::forward<Args>(args)...);
    return;
  }
  kernel<<<g, b, lds, stream>>>(static_cast<KArgs>(args)...);
}
}  // namespace pfa
// (hip_runtime.h defines hipLaunchKernelGGL as a macro; inside this library's device translation units every use of it
//  goes through pfa::launch_kernel)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
  ::pfa::launch_kernel(kernel, grid, block, lds, stream, __VA_ARGS__)

namespace pfa {

/// the same configuration with another cache policy (AUX)
template <typename Cfg, int AUX2>
struct with_aux;
template <typename T, typename Seq, int WG, int FPW, int PADS, int PADW, int TWM, int OCC, int AUX, int STAGED, int TWL,
          int AUX2>
struct with_aux<wg_cfg<T, Seq, WG, FPW, PADS, PADW, TWM, OCC, AUX, STAGED, TWL>, AUX2> {
  using type = wg_cfg<T, Seq, WG, FPW, PADS, PADW, TWM, OCC, AUX2, STAGED, TWL>;
};
template <typename Cfg, int AUX2>
using with_aux_t = typename with_aux<Cfg, AUX2>::type;

template <typename Cfg>
hipError_t launch_spec_prefetch(hipStream_t stream, unsigned grid, const void* in, void* out, const void* tw,
                                long long nfft, double scale, int backward) {
  using T = typename Cfg::T;
  const auto* i = static_cast<const cx<T>*>(in);
  auto* o = static_cast<cx<T>*>(out);
  const auto* t = static_cast<const cx<T>*>(tw);
  // two-tier grid for large launches: `grid` is the planner's uniform grid (k groups per work-group); three quarters
  // of the groups keep that shape, the last quarter goes to work-groups of 2 groups each
  const long long ngroups = (nfft + Cfg::FPW - 1) / Cfg::FPW;
  const long long k = grid > 0 ? (ngroups + grid - 1) / grid : 1;
  long long n_main = 0;
  static const bool uniform_only = getenv("PFFT_UNIFORM_GRID") != nullptr;  // A/B switch (profiles/r2_notes.md)
  if (k >= 4 && grid >= 4096 && !uniform_only) {
    n_main = (static_cast<long long>(grid) * 3 / 4) & ~255ll;
    const long long rest = ngroups - k * n_main;
    grid = static_cast<unsigned>(n_main + (rest + 1) / 2);
  }
  const int main_k = static_cast<int>(k);
  if (backward) {
    hipLaunchKernelGGL((stockham_wg_prefetch_kernel<Cfg, true>), dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, stream, i,
                       o, t, nfft, static_cast<T>(scale), n_main, main_k);
  } else {
    hipLaunchKernelGGL((stockham_wg_prefetch_kernel<Cfg, false>), dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, stream,
                       i, o, t, nfft, static_cast<T>(scale), n_main, main_k);
  }
  return hipGetLastError();
}

template <typename Cfg>
hipError_t launch_spec(hipStream_t stream, unsigned grid, const void* in, void* out, const void* tw, long long nfft,
                       double scale, int backward) {
  using T = typename Cfg::T;
  const auto* i = static_cast<const cx<T>*>(in);
  auto* o = static_cast<cx<T>*>(out);
  const auto* t = static_cast<const cx<T>*>(tw);
  if (backward) {
    hipLaunchKernelGGL((stockham_wg_kernel<Cfg, true>), dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, stream, i, o, t,
                       nfft, static_cast<T>(scale));
  } else {
    hipLaunchKernelGGL((stockham_wg_kernel<Cfg, false>), dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, stream, i, o, t,
                       nfft, static_cast<T>(scale));
  }
  return hipGetLastError();
}

template <typename Cfg>
hipError_t launch_spec_split(hipStream_t stream, unsigned grid, const void* in_re, const void* in_im, void* out_re,
                             void* out_im, const void* tw, long long nfft, double scale, int backward) {
  using T = typename Cfg::T;
  const auto* t = static_cast<const cx<T>*>(tw);
  const dim3 g(grid), b(Cfg::WG);
  if (backward) {
    hipLaunchKernelGGL((stockham_wg_split_kernel<Cfg, true>), g, b, Cfg::LDS_BYTES, stream, static_cast<const T*>(in_re),
                       static_cast<const T*>(in_im), static_cast<T*>(out_re), static_cast<T*>(out_im), t, nfft,
                       static_cast<T>(scale));
  } else {
    hipLaunchKernelGGL((stockham_wg_split_kernel<Cfg, false>), g, b, Cfg::LDS_BYTES, stream,
                       static_cast<const T*>(in_re), static_cast<const T*>(in_im), static_cast<T*>(out_re),
                       static_cast<T*>(out_im), t, nfft, static_cast<T>(scale));
  }
  return hipGetLastError();
}

template <typename Cfg>
spec_kernel make_spec_entry(int groups_per_wg = 1);

/// software-pipelined form (stockham_wg_prefetch_kernel) of a direct-I/O multi-pass variant
template <typename Cfg>
spec_kernel make_spec_entry_prefetch(int groups_per_wg = 4) {
  spec_kernel k = make_spec_entry<Cfg>(groups_per_wg);
  k.fn[0] = reinterpret_cast<const void*>(&stockham_wg_prefetch_kernel<Cfg, false>);
  k.fn[1] = reinterpret_cast<const void*>(&stockham_wg_prefetch_kernel<Cfg, true>);
  k.launch = &launch_spec_prefetch<Cfg>;
  return k;
}

/// the fields every form of a packed configuration shares (no kernel is instantiated by this)
template <typename Cfg>
spec_kernel spec_entry_fields(int groups_per_wg) {
  spec_kernel k{};
  k.groups_per_wg = groups_per_wg;
  k.precision = sizeof(typename Cfg::T) == 8 ? PFFT_PRECISION_F64 : PFFT_PRECISION_F32;
  k.n = Cfg::N;
  k.wg = Cfg::WG;
  k.fpw = Cfg::FPW;
  k.lds_bytes = Cfg::LDS_BYTES;
  k.n_radices = Cfg::NP;
  for (int i = 0; i < Cfg::NP; ++i) k.radices[i] = Cfg::Seq::r[i];
  k.tw_total = Cfg::Seq::tw_total;
  k.tw_in_regs = Cfg::TWM == TW_REGS ? 1 : 0;
  k.pads = Cfg::PADS;
  k.padw = Cfg::PADW;
  k.twm = Cfg::TWM;
  k.occ = Cfg::OCC;
  k.aux = Cfg::AUX;
  k.staged = Cfg::STAGED;
  k.twl = Cfg::TWL;
  return k;
}

template <typename Cfg>
spec_kernel make_spec_entry(int groups_per_wg) {
  spec_kernel k = spec_entry_fields<Cfg>(groups_per_wg);
  k.fn[0] = reinterpret_cast<const void*>(&stockham_wg_kernel<Cfg, false>);
  k.fn[1] = reinterpret_cast<const void*>(&stockham_wg_kernel<Cfg, true>);
  k.launch = &launch_spec<Cfg>;
  k.fn_split[0] = reinterpret_cast<const void*>(&stockham_wg_split_kernel<Cfg, false>);
  k.fn_split[1] = reinterpret_cast<const void*>(&stockham_wg_split_kernel<Cfg, true>);
  k.launch_split = &launch_spec_split<Cfg>;
  return k;
}

template <typename Cfg>
hipError_t launch_spec_xlane(hipStream_t stream, unsigned grid, const void* in, void* out, const void* tw,
                             long long nfft, double scale, int backward) {
  using T = typename Cfg::T;
  const auto* i = static_cast<const cx<T>*>(in);
  auto* o = static_cast<cx<T>*>(out);
  const auto* t = static_cast<const cx<T>*>(tw);
  if (backward) {
    hipLaunchKernelGGL((stockham_wg_xlane_kernel<Cfg, true>), dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, stream, i, o,
                       t, nfft, static_cast<T>(scale));
  } else {
    hipLaunchKernelGGL((stockham_wg_xlane_kernel<Cfg, false>), dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, stream, i, o,
                       t, nfft, static_cast<T>(scale));
  }
  return hipGetLastError();
}

/// cross-lane form (stockham_xlane.hpp) of an N = R * R staged variant; interleaved storage (split storage keeps
/// the LDS form of the same configuration)
template <typename Cfg>
spec_kernel make_spec_entry_xlane(int groups_per_wg = 1) {
  spec_kernel k = make_spec_entry<Cfg>(groups_per_wg);
  k.fn[0] = reinterpret_cast<const void*>(&stockham_wg_xlane_kernel<Cfg, false>);
  k.fn[1] = reinterpret_cast<const void*>(&stockham_wg_xlane_kernel<Cfg, true>);
  k.launch = &launch_spec_xlane<Cfg>;
  k.xlane = 1;
  return k;
}

template <typename Cfg, bool PF = false>
hipError_t launch_spec_hx(hipStream_t stream, unsigned grid, const void* in, void* out, const void* tw, long long nfft,
                          double scale, int backward) {
  using T = typename Cfg::T;
  const auto* i = static_cast<const cx<T>*>(in);
  auto* o = static_cast<cx<T>*>(out);
  const auto* t = static_cast<const cx<T>*>(tw);
  constexpr size_t lds = wg_hx_lds_bytes<Cfg>();
  if (backward) {
    hipLaunchKernelGGL((stockham_wg_hx_kernel<Cfg, true, PF>), dim3(grid), dim3(Cfg::WG), lds, stream, i, o, t, nfft, static_cast<T>(scale));
  } else {
    hipLaunchKernelGGL((stockham_wg_hx_kernel<Cfg, false, PF>), dim3(grid), dim3(Cfg::WG), lds, stream, i, o, t, nfft, static_cast<T>(scale));
  }
  return hipGetLastError();
}

template <typename Cfg, bool PF = false>
hipError_t launch_spec_hx_split(hipStream_t stream, unsigned grid, const void* in_re, const void* in_im, void* out_re,
                                void* out_im, const void* tw, long long nfft, double scale, int backward) {
  using T = typename Cfg::T;
  const auto* t = static_cast<const cx<T>*>(tw);
  const dim3 g(grid), b(Cfg::WG);
  constexpr size_t lds = wg_hx_lds_bytes<Cfg>();
  if (backward) {
    hipLaunchKernelGGL((stockham_wg_hx_split_kernel<Cfg, true, PF>), g, b, lds, stream, static_cast<const T*>(in_re),
                       static_cast<const T*>(in_im), static_cast<T*>(out_re), static_cast<T*>(out_im), t, nfft, static_cast<T>(scale));
  } else {
    hipLaunchKernelGGL((stockham_wg_hx_split_kernel<Cfg, false, PF>), g, b, lds, stream, static_cast<const T*>(in_re),
                       static_cast<const T*>(in_im), static_cast<T*>(out_re), static_cast<T*>(out_im), t, nfft, static_cast<T>(scale));
  }
  return hipGetLastError();
}

/// register-resident form (stockham_wg_hx.hpp) of a packed length; PF: its software-pipelined form
template <typename Cfg, bool PF = false>
spec_kernel make_spec_entry_hx(int groups_per_wg = 0) {
  // (fields only: the LDS-resident stockham_wg kernels of such a configuration -- fp32 32768 would need a 256 KiB image --
  //  are never instantiated, and every pointer of the entry is the register-resident kernel's: ADVICE r5)
  spec_kernel k = spec_entry_fields<Cfg>(groups_per_wg);
  k.lds_bytes = wg_hx_lds_bytes<Cfg>();
  k.fn[0] = reinterpret_cast<const void*>(&stockham_wg_hx_kernel<Cfg, false, PF>);
  k.fn[1] = reinterpret_cast<const void*>(&stockham_wg_hx_kernel<Cfg, true, PF>);
  k.launch = &launch_spec_hx<Cfg, PF>;
  k.fn_split[0] = reinterpret_cast<const void*>(&stockham_wg_hx_split_kernel<Cfg, false, PF>);
  k.fn_split[1] = reinterpret_cast<const void*>(&stockham_wg_hx_split_kernel<Cfg, true, PF>);
  k.launch_split = &launch_spec_hx_split<Cfg, PF>;
  k.hx = 1;
  return k;
}

/// Launch with one by-value argument struct.  args.any_order: the packet carries no barrier bit
/// (hipExtAnyOrderLaunch), so the work-groups may start while the previous launch of the stream is still draining;
/// the plan sets it only for launches that are independent of everything that can still be in flight (plan_exec.cpp,
/// chunk overlap).
template <typename K, typename A>
inline hipError_t pfa_launch(K kernel, dim3 g, dim3 b, size_t lds, hipStream_t stream, const A& args) {
  if (args.any_order != 0) {
    A copy = args;
    void* p[] = {&copy};
    return hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), g, b, p, lds, stream, nullptr, take_stop_event(),
                              hipExtAnyOrderLaunch);
  }
  hipLaunchKernelGGL(kernel, g, b, lds, stream, args);
  return hipGetLastError();
}

/// LDS behind the kernel's own for the store-modifier tables (strided_args::stw_tab)
template <typename Cfg>
inline size_t stw_lds_bytes(const strided_args& args, int stw) {
  return stw != 0 ? (static_cast<size_t>(args.stw_levels) << args.stw_lshift) * sizeof(cx<typename Cfg::T>) : 0;
}

template <typename Cfg>
hipError_t launch_strided(hipStream_t stream, unsigned grid, const strided_args& args, int backward, int stw) {
  const size_t lds = strided_lds_bytes<Cfg>() + stw_lds_bytes<Cfg>(args, stw);
  const dim3 g(grid), b(Cfg::WG);
  if (backward) {
    if (stw) {
      return pfa_launch(&stockham_strided_kernel<Cfg, true, true>, g, b, lds, stream, args);
    } else {
      return pfa_launch(&stockham_strided_kernel<Cfg, true, false>, g, b, lds, stream, args);
    }
  } else {
    if (stw) {
      return pfa_launch(&stockham_strided_kernel<Cfg, false, true>, g, b, lds, stream, args);
    } else {
      return pfa_launch(&stockham_strided_kernel<Cfg, false, false>, g, b, lds, stream, args);
    }
  }
  return hipGetLastError();
}

template <typename Cfg>
hipError_t launch_strided_prefetch(hipStream_t stream, unsigned grid, const strided_args& args, int backward, int stw) {
  const size_t lds = strided_lds_bytes<Cfg>() + stw_lds_bytes<Cfg>(args, stw);
  const dim3 g(grid), b(Cfg::WG);
  if (backward) {
    if (stw) {
      return pfa_launch(&stockham_strided_prefetch_kernel<Cfg, true, true>, g, b, lds, stream, args);
    } else {
      return pfa_launch(&stockham_strided_prefetch_kernel<Cfg, true, false>, g, b, lds, stream, args);
    }
  } else {
    if (stw) {
      return pfa_launch(&stockham_strided_prefetch_kernel<Cfg, false, true>, g, b, lds, stream, args);
    } else {
      return pfa_launch(&stockham_strided_prefetch_kernel<Cfg, false, false>, g, b, lds, stream, args);
    }
  }
  return hipGetLastError();
}

template <typename Cfg>
hipError_t launch_strided_row(hipStream_t stream, unsigned grid, const strided_args& args, int backward, int row_out) {
  constexpr size_t lds = strided_row_lds_bytes<Cfg>();
  const dim3 g(grid), b(Cfg::WG);
  if (row_out) {
    if (backward) {
      hipLaunchKernelGGL((stockham_strided_row_kernel<Cfg, true, false, true>), g, b, lds, stream, args);
    } else {
      hipLaunchKernelGGL((stockham_strided_row_kernel<Cfg, false, false, true>), g, b, lds, stream, args);
    }
  } else {
    if (backward) {
      hipLaunchKernelGGL((stockham_strided_row_kernel<Cfg, true, true, false>), g, b, lds, stream, args);
    } else {
      hipLaunchKernelGGL((stockham_strided_row_kernel<Cfg, false, true, false>), g, b, lds, stream, args);
    }
  }
  return hipGetLastError();
}

template <typename Cfg>
hipError_t launch_rows2d(hipStream_t stream, unsigned grid, const rows2d_args& args, int backward) {
  if (backward) {
    return pfa_launch(&stockham_rows2d_kernel<Cfg, true>, dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, stream, args);
  } else {
    return pfa_launch(&stockham_rows2d_kernel<Cfg, false>, dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, stream, args);
  }
  return hipGetLastError();
}

template <typename Cfg>
hipError_t launch_rows2d_split(hipStream_t stream, unsigned grid, const rows2d_args& args, int backward) {
  if (backward) {
    hipLaunchKernelGGL((stockham_rows2d_kernel<Cfg, true, true>), dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, stream, args);
  } else {
    hipLaunchKernelGGL((stockham_rows2d_kernel<Cfg, false, true>), dim3(grid), dim3(Cfg::WG), Cfg::LDS_BYTES, stream, args);
  }
  return hipGetLastError();
}

template <typename Cfg>
rows2d_kernel make_rows2d_entry(int groups_per_wg);

/// the entry for Cfg (cache policy nt) and its writer twin (default-policy stores)
template <typename Cfg>
void add_rows2d_entries(std::vector<rows2d_kernel>& v, int groups_per_wg) {
  rows2d_kernel k = make_rows2d_entry<Cfg>(groups_per_wg);
  k.fn_split[0] = reinterpret_cast<const void*>(&stockham_rows2d_kernel<Cfg, false, true>);
  k.fn_split[1] = reinterpret_cast<const void*>(&stockham_rows2d_kernel<Cfg, true, true>);
  k.launch_split = &launch_rows2d_split<Cfg>;
  v.push_back(k);
  rows2d_kernel w = make_rows2d_entry<with_aux_t<Cfg, PFA_AUX_WRITER>>(groups_per_wg);
  w.policy = 1;
  // (round 6: the split-storage form of the writer twin too -- the two-pass 2-D plan of SPLIT_COMPLEX data in cache-sized chunks)
  using WCfg = with_aux_t<Cfg, PFA_AUX_WRITER>;
  w.fn_split[0] = reinterpret_cast<const void*>(&stockham_rows2d_kernel<WCfg, false, true>);
  w.fn_split[1] = reinterpret_cast<const void*>(&stockham_rows2d_kernel<WCfg, true, true>);
  w.launch_split = &launch_rows2d_split<WCfg>;
  v.push_back(w);
}

/// Cfg: the row FFT's wg_cfg with FPW = rows per work-group (= the column radix)
template <typename Cfg>
rows2d_kernel make_rows2d_entry(int groups_per_wg) {
  rows2d_kernel k{};
  k.precision = sizeof(typename Cfg::T) == 8 ? PFFT_PRECISION_F64 : PFFT_PRECISION_F32;
  k.n = Cfg::N;
  k.rc = Cfg::FPW;
  k.wg = Cfg::WG;
  k.lds_bytes = Cfg::LDS_BYTES;
  k.n_radices = Cfg::NP;
  for (int i = 0; i < Cfg::NP; ++i) k.radices[i] = Cfg::Seq::r[i];
  k.groups_per_wg = groups_per_wg;
  k.fn[0] = reinterpret_cast<const void*>(&stockham_rows2d_kernel<Cfg, false>);
  k.fn[1] = reinterpret_cast<const void*>(&stockham_rows2d_kernel<Cfg, true>);
  k.launch = &launch_rows2d<Cfg>;
  return k;
}

/// mark an entry as the wide-group alternative of its length (strided_kernel::wide)
inline strided_kernel wide(strided_kernel k) {
  k.wide = 1;
  return k;
}

/// mark an entry as the alternative of its length for stages with a row-shaped side (strided_kernel::rowish)
inline strided_kernel rowish(strided_kernel k) {
  k.rowish = 1;
  return k;
}

/// add the row-staged forms to an entry (fp32: a wave covers only 64/FPW * 8 B of a row when addressed f-fastest)
template <typename Cfg>
strided_kernel with_rows(strided_kernel k) {
  k.fn_row[0] = reinterpret_cast<const void*>(&stockham_strided_row_kernel<Cfg, false, true, false>);
  k.fn_row[1] = reinterpret_cast<const void*>(&stockham_strided_row_kernel<Cfg, true, true, false>);
  k.fn_row[2] = reinterpret_cast<const void*>(&stockham_strided_row_kernel<Cfg, false, false, true>);
  k.fn_row[3] = reinterpret_cast<const void*>(&stockham_strided_row_kernel<Cfg, true, false, true>);
  k.lds_bytes_row = strided_row_lds_bytes<Cfg>();
  k.launch_row = &launch_strided_row<Cfg>;
  return k;
}

/// (PF: the software-pipelined kernel's tiled-input form; LTW: ... carrying the inter-stage twiddles on its loads, tables
/// in LDS behind the kernel's own)
template <typename Cfg, bool PF = false, int LTW = 0>
hipError_t launch_strided_tin(hipStream_t stream, unsigned grid, const strided_args& args, int backward) {
  const size_t lds = strided_lds_bytes<Cfg>() + stw_lds_bytes<Cfg>(args, LTW);
  const dim3 g(grid), b(Cfg::WG);
  if constexpr (PF) {
    if (backward) return pfa_launch(&stockham_strided_prefetch_kernel<Cfg, true, 0, 0, true, LTW>, g, b, lds, stream, args);
    return pfa_launch(&stockham_strided_prefetch_kernel<Cfg, false, 0, 0, true, LTW>, g, b, lds, stream, args);
  } else {
    static_assert(LTW == 0, "the load-side modifier exists in the software-pipelined kernel only");
    if (backward) return pfa_launch(&stockham_strided_kernel<Cfg, true, 0, 0, true>, g, b, lds, stream, args);
    return pfa_launch(&stockham_strided_kernel<Cfg, false, 0, 0, true>, g, b, lds, stream, args);
  }
}

/// add the tiled-input form (four-step stage B reading a group-major intermediate) to an entry
template <typename Cfg, bool PF = false, int LTW = 0>
strided_kernel with_tin(strided_kernel k) {
  static_assert(tin_supported<Cfg>(), "see tin_supported()");
  if constexpr (PF) {
    k.fn_tin[0] = reinterpret_cast<const void*>(&stockham_strided_prefetch_kernel<Cfg, false, 0, 0, true, LTW>);
    k.fn_tin[1] = reinterpret_cast<const void*>(&stockham_strided_prefetch_kernel<Cfg, true, 0, 0, true, LTW>);
  } else {
    k.fn_tin[0] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, false, 0, 0, true>);
    k.fn_tin[1] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, true, 0, 0, true>);
  }
  k.launch_tin = &launch_strided_tin<Cfg, PF, LTW>;
  return k;
}

/// tiled-input form for tiles twice as wide as the entry's groups (strided_kernel::launch_tin_w)
template <typename Cfg>
hipError_t launch_strided_tin_w(hipStream_t stream, unsigned grid, const strided_args& args, int backward) {
  const size_t lds = strided_lds_bytes<Cfg>();
  const dim3 g(grid), b(Cfg::WG);
  if (backward) return pfa_launch(&stockham_strided_kernel<Cfg, true, 0, 0, 2 * Cfg::FPW>, g, b, lds, stream, args);
  return pfa_launch(&stockham_strided_kernel<Cfg, false, 0, 0, 2 * Cfg::FPW>, g, b, lds, stream, args);
}

template <typename Cfg>
strided_kernel with_tin_w(strided_kernel k) {
  static_assert(tin_supported<Cfg, 2 * Cfg::FPW>(), "see tin_supported()");
  k.fn_tin_w[0] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, false, 0, 0, 2 * Cfg::FPW>);
  k.fn_tin_w[1] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, true, 0, 0, 2 * Cfg::FPW>);
  k.launch_tin_w = &launch_strided_tin_w<Cfg>;
  k.tin_w = 2 * Cfg::FPW;
  return k;
}

template <typename Cfg>
hipError_t launch_strided_split(hipStream_t stream, unsigned grid, const strided_args& args, int backward) {
  constexpr size_t lds = strided_lds_bytes<Cfg>();
  if (backward) {
    hipLaunchKernelGGL((stockham_strided_kernel<Cfg, true, false, true>), dim3(grid), dim3(Cfg::WG), lds, stream, args);
  } else {
    hipLaunchKernelGGL((stockham_strided_kernel<Cfg, false, false, true>), dim3(grid), dim3(Cfg::WG), lds, stream, args);
  }
  return hipGetLastError();
}

template <typename Cfg>
strided_kernel make_strided_entry(int groups_per_wg = 1);

template <typename Cfg>
strided_kernel make_strided_entry_prefetch(int groups_per_wg = 4) {
  strided_kernel k = make_strided_entry<Cfg>(groups_per_wg);
  k.fn[0] = reinterpret_cast<const void*>(&stockham_strided_prefetch_kernel<Cfg, false, false>);
  k.fn[1] = reinterpret_cast<const void*>(&stockham_strided_prefetch_kernel<Cfg, false, true>);
  k.fn[2] = reinterpret_cast<const void*>(&stockham_strided_prefetch_kernel<Cfg, true, false>);
  k.fn[3] = reinterpret_cast<const void*>(&stockham_strided_prefetch_kernel<Cfg, true, true>);
  k.launch = &launch_strided_prefetch<Cfg>;
  return k;
}

/// flags of add_strided_entries
/// SE_FS_A / SE_FS_B: the entry of its length for the four-step stage A / stage B (strided_kernel::fs_a / fs_b)
enum : unsigned { SE_ROWS = 1, SE_TIN = 2, SE_WIDE = 4, SE_ROWISH = 8, SE_PREFETCH = 16, SE_FS_A = 32, SE_FS_B = 64, SE_FS_ONLY = 128,
                SE_LTW = 256 /* fs_b entry carrying the modifier on its loads (strided_kernel::fs_ltw) */,
                SE_PLAIN_WRITER = 512 /* the writer twin also carries the forms without store modifier (stage A of an SE_LTW pair) */,
                SE_TIN_W = 1024 /* ... and the tiled-input form for tiles of 2 * FPW elements (strided_kernel::launch_tin_w) */ };

template <typename Cfg, unsigned F>
strided_kernel make_strided_entry_flags(int groups_per_wg) {
  strided_kernel k = (F & SE_PREFETCH) ? make_strided_entry_prefetch<Cfg>(groups_per_wg) : make_strided_entry<Cfg>(groups_per_wg);
  if constexpr ((F & SE_ROWS) != 0) k = with_rows<Cfg>(k);
  if constexpr ((F & SE_TIN) != 0) k = with_tin<Cfg, (F & SE_PREFETCH) != 0, (F & SE_LTW) != 0 ? 1 : 0>(k);
  if constexpr ((F & SE_TIN_W) != 0) k = with_tin_w<Cfg>(k);
  k.fs_ltw = (F & SE_LTW) != 0;
  k.wide = (F & SE_WIDE) != 0;
  k.rowish = (F & SE_ROWISH) != 0;
  k.fs_a = (F & SE_FS_A) != 0;
  k.fs_b = (F & SE_FS_B) != 0;
  k.fs_only = (F & SE_FS_ONLY) != 0;
  static_assert((F & SE_FS_B) == 0 || (F & SE_TIN) != 0, "a four-step stage-B entry needs its tiled-input form");
  return k;
}

/// launchers of the policy twins: a writer only exists with the store modifier (four-step stage A), a reader only
/// without it (stage B, second pass of the two-pass 2-D plan)
template <typename Cfg, bool PREFETCH>
hipError_t launch_strided_writer(hipStream_t stream, unsigned grid, const strided_args& args, int backward, int stw) {
  if (!stw) return hipErrorInvalidValue;
  const size_t lds = strided_lds_bytes<Cfg>() + stw_lds_bytes<Cfg>(args, stw);
  const dim3 g(grid), b(Cfg::WG);
  if constexpr (PREFETCH) {
    if (backward) return pfa_launch(&stockham_strided_prefetch_kernel<Cfg, true, true>, g, b, lds, stream, args);
    else return pfa_launch(&stockham_strided_prefetch_kernel<Cfg, false, true>, g, b, lds, stream, args);
  } else {
    if (backward) return pfa_launch(&stockham_strided_kernel<Cfg, true, true>, g, b, lds, stream, args);
    else return pfa_launch(&stockham_strided_kernel<Cfg, false, true>, g, b, lds, stream, args);
  }
  return hipGetLastError();
}
template <typename Cfg, bool PREFETCH>
hipError_t launch_strided_reader(hipStream_t stream, unsigned grid, const strided_args& args, int backward, int stw) {
  if (stw) return hipErrorInvalidValue;
  constexpr size_t lds = strided_lds_bytes<Cfg>();
  const dim3 g(grid), b(Cfg::WG);
  if constexpr (PREFETCH) {
    if (backward) return pfa_launch(&stockham_strided_prefetch_kernel<Cfg, true, false>, g, b, lds, stream, args);
    else return pfa_launch(&stockham_strided_prefetch_kernel<Cfg, false, false>, g, b, lds, stream, args);
  } else {
    if (backward) return pfa_launch(&stockham_strided_kernel<Cfg, true, false>, g, b, lds, stream, args);
    else return pfa_launch(&stockham_strided_kernel<Cfg, false, false>, g, b, lds, stream, args);
  }
  return hipGetLastError();
}
template <typename Cfg>
hipError_t launch_strided_row_in(hipStream_t stream, unsigned grid, const strided_args& args, int backward, int row_out) {
  if (row_out) return hipErrorInvalidValue;
  constexpr size_t lds = strided_row_lds_bytes<Cfg>();
  const dim3 g(grid), b(Cfg::WG);
  if (backward) hipLaunchKernelGGL((stockham_strided_row_kernel<Cfg, true, true, false>), g, b, lds, stream, args);
  else hipLaunchKernelGGL((stockham_strided_row_kernel<Cfg, false, true, false>), g, b, lds, stream, args);
  return hipGetLastError();
}

/// policy twin of `base` (the nt entry of the same shape): Cfg carries the twin's AUX; only the forms a writer
/// (policy 1) or a reader (policy 2) is ever launched in are instantiated, the others stay null
template <typename Cfg, unsigned F>
strided_kernel make_strided_twin(const strided_kernel& base, int policy) {
  constexpr bool PF = (F & SE_PREFETCH) != 0;
  strided_kernel k = base;
  k.policy = policy;
  for (auto& fp : k.fn) fp = nullptr;
  for (auto& fp : k.fn_row) fp = nullptr;
  k.fn_split[0] = k.fn_split[1] = nullptr;
  k.fn_tin[0] = k.fn_tin[1] = nullptr;
  k.fn_tin_w[0] = k.fn_tin_w[1] = nullptr;
  k.launch_tin_w = nullptr;
  k.tin_w = 0;
  k.launch_split = nullptr;
  k.launch_row = nullptr;
  k.launch_tin = nullptr;
  if (policy == 1) {
    if constexpr (PF) {
      k.fn[1] = reinterpret_cast<const void*>(&stockham_strided_prefetch_kernel<Cfg, false, true>);
      k.fn[3] = reinterpret_cast<const void*>(&stockham_strided_prefetch_kernel<Cfg, true, true>);
    } else {
      k.fn[1] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, false, true>);
      k.fn[3] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, true, true>);
    }
    k.launch = &launch_strided_writer<Cfg, PF>;
    if constexpr ((F & SE_PLAIN_WRITER) != 0) {  // stage A of a pair whose stage B carries the modifier: no store modifier
      static_assert(!PF, "plain writer forms: the non-pipelined kernel");
      k.fn[0] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, false, false>);
      k.fn[2] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, true, false>);
      k.launch = &launch_strided<Cfg>;
    }
  } else {
    if constexpr (PF) {
      k.fn[0] = reinterpret_cast<const void*>(&stockham_strided_prefetch_kernel<Cfg, false, false>);
      k.fn[2] = reinterpret_cast<const void*>(&stockham_strided_prefetch_kernel<Cfg, true, false>);
    } else {
      k.fn[0] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, false, false>);
      k.fn[2] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, true, false>);
    }
    k.launch = &launch_strided_reader<Cfg, PF>;
    if constexpr ((F & SE_ROWS) != 0) {
      k.fn_row[0] = reinterpret_cast<const void*>(&stockham_strided_row_kernel<Cfg, false, true, false>);
      k.fn_row[1] = reinterpret_cast<const void*>(&stockham_strided_row_kernel<Cfg, true, true, false>);
      k.launch_row = &launch_strided_row_in<Cfg>;
    }
    if constexpr ((F & SE_TIN) != 0) k = with_tin<Cfg, PF, (F & SE_LTW) != 0 ? 1 : 0>(k);
    if constexpr ((F & SE_TIN_W) != 0) k = with_tin_w<Cfg>(k);
    // (round 6: the reader twin of a wide entry -- the second pass of the two-pass 2-D plan -- carries its split-storage form)
    if constexpr ((F & SE_WIDE) != 0 && !PF) {
      k.fn_split[0] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, false, false, true>);
      k.fn_split[1] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, true, false, true>);
      k.launch_split = &launch_strided_split<Cfg>;
    }
  }
  return k;
}

/// the entry for Cfg (cache policy nt) and its writer / reader twins
template <typename Cfg, unsigned F = 0>
void add_strided_entries(std::vector<strided_kernel>& v, int groups_per_wg = 1, int fs_groups_per_wg = 0) {
  strided_kernel base = make_strided_entry_flags<Cfg, F>(groups_per_wg);
  base.fs_groups_per_wg = fs_groups_per_wg;
  v.push_back(base);
  v.push_back(make_strided_twin<with_aux_t<Cfg, PFA_AUX_WRITER>, F>(base, 1));
  v.push_back(make_strided_twin<with_aux_t<Cfg, PFA_AUX_READER>, F>(base, 2));
}

template <typename Cfg>
strided_kernel make_strided_entry(int groups_per_wg) {
  strided_kernel k{};
  k.groups_per_wg = groups_per_wg;
  k.precision = sizeof(typename Cfg::T) == 8 ? PFFT_PRECISION_F64 : PFFT_PRECISION_F32;
  k.n = Cfg::N;
  k.wg = Cfg::WG;
  k.fpw = Cfg::FPW;
  k.lds_bytes = strided_lds_bytes<Cfg>();
  k.stw_mode = 1;
  k.n_radices = Cfg::NP;
  for (int i = 0; i < Cfg::NP; ++i) k.radices[i] = Cfg::Seq::r[i];
  k.fn[0] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, false, false>);
  k.fn[1] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, false, true>);
  k.fn[2] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, true, false>);
  k.fn[3] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, true, true>);
  k.launch = &launch_strided<Cfg>;
  k.fn_split[0] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, false, false, true>);
  k.fn_split[1] = reinterpret_cast<const void*>(&stockham_strided_kernel<Cfg, true, false, true>);
  k.launch_split = &launch_strided_split<Cfg>;
  return k;
}

}  // namespace pfa
