// XCD-local four-step kernel instantiations for gfx950 (stockham_xcd.hpp): both stages of N = n1 x n2 in one persistent
// launch.  Registered only where the launch beats the two-launch plan of the same stage bodies on hardware
// (tools/tune_xcd.hip, profiles/r4_xcd_local.md):
//   fp32 512 x 512 (N = 2^18): 0.43-0.44 of the HBM peak against 0.35 (the 512 x 512 pair) / 0.365 (256 x 1024, the split
//   the two-launch planner takes) -- 8.8.8 on 512 lanes x 16 columns, 64 KiB tasks, two work-groups per CU.
// Measured and NOT registered: fp32 256 x 256 (0.29-0.31 against 0.38: 32 KiB tasks, the hand-off bookkeeping of a task is
// a quarter of its time and a sync-free run of the same loop only ties the two launches), fp64 256 x 256 (0.31-0.33
// against 0.34), fp64 512 x 512 and 1024 x 1024 (one work-group per CU: 0.29-0.30 against 0.33).
#include "kernels_impl.hpp"
#include "stockham_xcd.hpp"

namespace pfa {

namespace {

template <typename Cfg, int OCCX>
hipError_t launch_xcd(hipStream_t stream, unsigned grid, size_t lds, const xcd_args& args, int backward) {
  if (backward) {
    hipLaunchKernelGGL((stockham_xcd_fourstep_kernel<Cfg, Cfg, true, 1, 1, 0, OCCX>), dim3(grid), dim3(Cfg::WG), lds, stream, args);
  } else {
    hipLaunchKernelGGL((stockham_xcd_fourstep_kernel<Cfg, Cfg, false, 1, 1, 0, OCCX>), dim3(grid), dim3(Cfg::WG), lds, stream, args);
  }
  return hipGetLastError();
}

/// square pair on one configuration; OCCX: waves per SIMD the register budget leaves room for (work-groups per CU x
/// waves per work-group / 4)
template <typename Cfg, int OCCX>
xcd_kernel make_xcd_entry(int slots, int lag, int lookahead) {
  xcd_kernel k{};
  k.precision = sizeof(typename Cfg::T) == 8 ? PFFT_PRECISION_F64 : PFFT_PRECISION_F32;
  k.n1 = Cfg::N;
  k.n2 = Cfg::N;
  k.wg = Cfg::WG;
  k.fpw = Cfg::FPW;
  k.lds_bytes = strided_lds_bytes<Cfg>();
  k.n_radices = Cfg::NP;
  for (int i = 0; i < Cfg::NP; ++i) k.radices[i] = Cfg::Seq::r[i];
  k.fn[0] = reinterpret_cast<const void*>(&stockham_xcd_fourstep_kernel<Cfg, Cfg, false, 1, 1, 0, OCCX>);
  k.fn[1] = reinterpret_cast<const void*>(&stockham_xcd_fourstep_kernel<Cfg, Cfg, true, 1, 1, 0, OCCX>);
  k.launch = &launch_xcd<Cfg, OCCX>;
  k.slots = slots;
  k.lag = lag;
  k.lookahead = lookahead;
  return k;
}

std::vector<xcd_kernel> build() {
  std::vector<xcd_kernel> v;
  // slots / lag: tools/tune_xcd.hip case 18 -- (12, 8) 1.22-1.26 ms per 1024 transforms, (8, 5) 1.24-1.29, (6, 4) 1.31-1.34,
  // (4, 2) 1.39-1.50 (stage-B tasks wait for their input), lag = slots - 1 2.0-2.2 (stage-A tasks wait for their slot)
  v.push_back(make_xcd_entry<strided_cfg<float, radix_list<8, 8, 8>, 512, 16, 2, PFA_AUX_NT>, 4>(12, 8, 4));
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
