// XCD-local four-step kernel instantiations for gfx950 (stockham_xcd.hpp): both stages of N = n1 x n2 in one persistent
// launch.  Registered only where the launch beats the two-launch plan of the same stage bodies on hardware
// (tools/tune_xcd.hip sweeps; profiles/r4_xcd_local.md has the table): of the HBM peak, single launch against the two
// launches of the same pair --
//   fp32  2^16 256 x 256   0.41 / 0.38      fp64  2^16 256 x 256   0.46 / 0.34
//         2^17 256 x 512   0.42 / 0.37            2^17 256 x 512   0.43 / 0.33
//         2^18 512 x 512   0.42-0.44 / 0.35       2^18 512 x 512   0.41 / 0.32
//         2^19 1024 x 512  0.38 / 0.33 (*)        2^19 512 x 1024  0.385 / 0.35 (*)
//         2^20 1024 x 1024 0.38 / 0.32            (*) not registered since round 5: +1...4 % through the library
// and through the library against the plan the two-launch planner takes (its own split, chunked to the Infinity Cache;
// tools/probes/xcd_lib_ab*.sh): +6...12 % on fp32 2^16 / 2^17 / 2^18 / 2^20 and fp64 2^16 / 2^17 / 2^18 from 1 GiB of
// data up, +1...4 % on the two 2^19 pairs (hence not registered); below 0.5 GiB (0.75 GiB for fp32 2^16) the two launches win -- the
// persistent launch has a fixed start-up of 20-30 us and the two-launch plan's data starts to fit the Infinity Cache.
// Measured and NOT registered: fp32 2^15 (128 x 256: 0.34 against 0.38 -- 16...32 KiB tasks, the hand-off bookkeeping of
// a task is a quarter of its time), fp64 2^20 (1024 x 1024, BASELINE configs[2]: 0.34-0.355 against 0.365).
#include "kernels_impl.hpp"
#include "stockham_xcd.hpp"

namespace pfa {

namespace {

template <typename CfgA, typename CfgB, int OCCX, int WG>
hipError_t launch_xcd(hipStream_t stream, unsigned grid, size_t lds, const xcd_args& args, int backward) {
  if (backward) {
    hipLaunchKernelGGL((stockham_xcd_fourstep_kernel<CfgA, CfgB, true, 1, 1, 0, OCCX, WG>), dim3(grid), dim3(WG), lds, stream, args);
  } else {
    hipLaunchKernelGGL((stockham_xcd_fourstep_kernel<CfgA, CfgB, false, 1, 1, 0, OCCX, WG>), dim3(grid), dim3(WG), lds, stream, args);
  }
  return hipGetLastError();
}

/// the recovery launch behind it (stockham_xcd_recover_kernel; `phase`: xcd_args.hpp)
template <typename CfgA, typename CfgB, int OCCX, int WG>
hipError_t launch_xcd_recover(hipStream_t stream, unsigned grid, size_t lds, const xcd_args& args, int backward, int phase) {
  if (backward) {
    hipLaunchKernelGGL((stockham_xcd_recover_kernel<CfgA, CfgB, true, 1, 1, OCCX, WG>), dim3(grid), dim3(WG), lds, stream, args, phase);
  } else {
    hipLaunchKernelGGL((stockham_xcd_recover_kernel<CfgA, CfgB, false, 1, 1, OCCX, WG>), dim3(grid), dim3(WG), lds, stream, args, phase);
  }
  return hipGetLastError();
}

/// CfgA x CfgB on work-groups of WG lanes; OCCX: waves per SIMD the register budget leaves room for (work-groups per
/// CU x waves per work-group / 4)
template <typename CfgA, typename CfgB, int OCCX, int WG = (CfgA::WG > CfgB::WG ? CfgA::WG : CfgB::WG)>
xcd_kernel make_xcd_entry(int slots, int lag, int lookahead, int min_mib = 512, int wg_per_cu = 0) {
  using L = xcd_layout<CfgA, CfgB, WG>;
  using T = typename CfgA::T;
  xcd_kernel k{};
  k.precision = sizeof(T) == 8 ? PFFT_PRECISION_F64 : PFFT_PRECISION_F32;
  k.n1 = CfgA::N;
  k.n2 = CfgB::N;
  k.wg = WG;
  k.fpw = CfgA::FPW;
  k.tasks_a = CfgB::N / CfgA::FPW / L::HA;
  k.tasks_b = CfgA::N / CfgB::FPW / L::HB;
  k.twl_a_off = static_cast<unsigned>(L::TWL_A * sizeof(cx<T>));
  k.twl_b_off = static_cast<unsigned>(L::TWL_B * sizeof(cx<T>));
  k.stw_off = static_cast<unsigned>(L::STW * sizeof(cx<T>));
  k.n_radices_a = CfgA::NP;
  k.n_radices_b = CfgB::NP;
  for (int i = 0; i < CfgA::NP; ++i) k.radices_a[i] = CfgA::Seq::r[i];
  for (int i = 0; i < CfgB::NP; ++i) k.radices_b[i] = CfgB::Seq::r[i];
  k.fn[0] = reinterpret_cast<const void*>(&stockham_xcd_fourstep_kernel<CfgA, CfgB, false, 1, 1, 0, OCCX, WG>);
  k.fn[1] = reinterpret_cast<const void*>(&stockham_xcd_fourstep_kernel<CfgA, CfgB, true, 1, 1, 0, OCCX, WG>);
  k.launch = &launch_xcd<CfgA, CfgB, OCCX, WG>;
  k.fn_recover[0] = reinterpret_cast<const void*>(&stockham_xcd_recover_kernel<CfgA, CfgB, false, 1, 1, OCCX, WG>);
  k.fn_recover[1] = reinterpret_cast<const void*>(&stockham_xcd_recover_kernel<CfgA, CfgB, true, 1, 1, OCCX, WG>);
  k.launch_recover = &launch_xcd_recover<CfgA, CfgB, OCCX, WG>;
  k.slots = slots;
  k.lag = lag;
  k.lookahead = lookahead;
  k.wg_per_cu = wg_per_cu;
  k.min_mib = min_mib;
  return k;
}

std::vector<xcd_kernel> build() {
  std::vector<xcd_kernel> v;
  // (slots, lag): tools/tune_xcd.hip TUNE_SWEEP=1.  Tasks of 32-64 KiB want deep rings (a queue's work-groups run up to
  // lag transforms ahead of the stage-B reads): fp32 2^18 (12, 8) 1.22-1.26 ms per 1024 transforms, (8, 5) 1.24-1.29,
  // (6, 4) 1.31-1.34, (4, 2) 1.39-1.50, lag = slots - 1 2.0-2.2 (stage-A tasks wait for their slot); tasks of 128 KiB
  // (one work-group per CU) are served by (4, 2)...(6, 4).
  using c256 = strided_cfg<float, radix_list<16, 16>, 256, 16, 2, PFA_AUX_NT>;
  using c512 = strided_cfg<float, radix_list<8, 8, 8>, 512, 16, 2, PFA_AUX_NT>;
  using c1024 = strided_cfg<float, radix_list<16, 8, 8>, 1024, 16, 4, PFA_AUX_NT>;
  v.push_back(make_xcd_entry<c256, c256, 4>(24, 12, 4, 768));  // four work-groups per CU
  v.push_back(make_xcd_entry<c256, c512, 4, 512>(24, 12, 4));  // two 256-lane stage-A groups side by side per task
  v.push_back(make_xcd_entry<c512, c512, 4>(12, 8, 4));
  // (fp32 2^19 as 1024 x 512 -- two 512-lane stage-B groups side by side, schedule (6, 4) -- and fp64 2^19 as 512 x 1024 were
  //  registered in round 4 for +1...4 % over the two-launch plan: inside the +-3 % box-to-box spread, and the recovery
  //  launch's bookkeeping costs the single launch 1-2 %.  De-registered in round 5; tools/tune_xcd.hip cases 191 / 119 keep them.)
  v.push_back(make_xcd_entry<c1024, c1024, 4>(4, 2, 4));
  using d256 = strided_cfg<double, radix_list<16, 16>, 128, 8, 2, PFA_AUX_NT>;
  using d512 = strided_cfg<double, radix_list<8, 8, 8>, 512, 8, 2, PFA_AUX_NT>;
  v.push_back(make_xcd_entry<d256, d256, 2, 512>(16, 8, 4));  // four 128-lane groups per task, one work-group per CU
  v.push_back(make_xcd_entry<d256, d512, 2, 512>(16, 8, 4));
  v.push_back(make_xcd_entry<d512, d512, 2, 1024>(4, 2, 4));  // two 512-lane groups per task
  return v;
}

__global__ void xcd_census_kernel(unsigned* out) {
  if (threadIdx.x == 0) atomicMax(out, xcd_id() + 1u);
}

}  // namespace

const xcd_kernel* xcd_kernels(int* count) {
  static const std::vector<xcd_kernel> g = build();
  *count = static_cast<int>(g.size());
  return g.data();
}

/// XCC ids the device hands to work-groups (highest id seen by a grid that covers every XCD many times over, + 1).
/// The kernel's queues are indexed by that id; an id beyond the census finds no queue and idles (stockham_xcd.hpp).
int xcd_census(hipStream_t stream) {
  unsigned* d = nullptr;
  if (hipMalloc(&d, sizeof(unsigned)) != hipSuccess) return 0;
  unsigned h = 0;
  if (hipMemsetAsync(d, 0, sizeof(unsigned), stream) == hipSuccess) {
    hipLaunchKernelGGL(xcd_census_kernel, dim3(4096), dim3(64), 0, stream, d);
    if (hipMemcpyAsync(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
      h = 0;
    }
  }
  (void)hipFree(d);
  return static_cast<int>(h);
}

}  // namespace pfa
