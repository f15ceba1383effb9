// Experiment (not adopted, profiles/r3_notes.md section 16): "double group" stage A -- 2 * FPW columns loaded as whole
// 128-byte lines, the second half's pass-0 outputs parked in registers.  Bit-identical to production; 49-79 VGPRs spilled
// at the 128-register budget of 1024 lanes (67 at 256 registers with 512 lanes), 190-213 us against 168.
#pragma once
#include "../../portfft_amd/csrc/stockham_strided.hpp"

namespace pfa {

/// "Double group" form of a four-step stage A whose LDS image holds only half a 128-byte line of columns (fp32 n = 2048:
/// 8 columns x 8 B): the work-group LOADS 2 * FPW adjacent columns with 2 * FPW lanes per row -- whole lines, like the
/// 16-column kernels --, runs pass 0 of both halves in registers, and takes the two halves through the remaining passes
/// one after the other (the image of FPW columns is all the LDS there is); the pass-0 outputs of the second half wait in
/// registers meanwhile.  Stores, store modifier and output addressing are those of stockham_strided_kernel on the
/// FPW-column groups 2g and 2g + 1, so the intermediate and its stage B do not change.  Needs inner % (2 * FPW) == 0
/// and (N / R0) % (WG / (2 * FPW)) == 0.  Same role: reference common/global.hpp:135-170.
template <typename Cfg, bool BWD, int STW>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_strided_dg_kernel(const strided_args a) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  using CfgW = wg_cfg<T, Seq, Cfg::WG, 2 * Cfg::FPW, 0, 0, Cfg::TWM, Cfg::OCC, Cfg::AUX, 0, Cfg::TWL>;
  constexpr int R0 = Seq::r[0];
  constexpr int BPT = CfgW::bpt(0);
  static_assert(Cfg::NP >= 2 && (Cfg::N / R0) % CfgW::TPF == 0, "double group: pass 0 must divide evenly over the lanes");
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_strided[];
  cx<T>* lds = reinterpret_cast<cx<T>*>(pfa_smem_strided);
  const unsigned f = threadIdx.x % Cfg::FPW, tid = threadIdx.x / Cfg::FPW;
  const unsigned fw = threadIdx.x % CfgW::FPW, tidw = threadIdx.x / CfgW::FPW;
  const unsigned half = fw / Cfg::FPW, fh = fw % Cfg::FPW;
  const cx<T>* __restrict__ tw = static_cast<const cx<T>*>(a.tw);
  const long long ngw = strided_ngroups<CfgW>(a);
  strided_copy_twiddles<Cfg>(lds, tw);
  strided_copy_stw<Cfg, STW>(a);
  for (long long gw = blockIdx.x; gw < ngw; gw += gridDim.x) {
    bool livew;
    long long c0w, nlivew;
    const auto iow = strided_group<CfgW, 0>(a, gw, fw, &livew, &c0w, &nlivew);
    cx<T> v[BPT][R0];
    strided_pass0_load<CfgW, BWD>(iow, a, fw, tidw, livew, v);
    sfor<0, BPT>([&](auto i_) PFA_LAMBDA { dft<R0>(v[decltype(i_)::value]); });
    sfor<0, 2>([&](auto h_) PFA_LAMBDA {
      constexpr unsigned h = decltype(h_)::value;
      // (the last pass of the previous half ends its LDS reads with a barrier before it stores: no barrier needed here)
      if (half == h) {
        sfor<0, BPT>([&](auto i_) PFA_LAMBDA {
          constexpr int i = decltype(i_)::value;
          const unsigned j = tidw + i * CfgW::TPF;
          cx<T>* p = lds + (j * R0) * Cfg::FPW + fh;
          sfor<0, R0>([&](auto u_) PFA_LAMBDA {
            constexpr int u = decltype(u_)::value;
            p[u * Cfg::FPW] = v[i][u];
          });
        });
      }
      __syncthreads();
      bool live;
      long long c0, nlive;
      const auto io = strided_group<Cfg, 0>(a, 2 * gw + h, f, &live, &c0, &nlive);
      strided_passes<Cfg, BWD, STW, 1, decltype(io)>(io, a, f, tid, live, c0, lds, tw, nlive);
    });
  }
}

}  // namespace pfa
