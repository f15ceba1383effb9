// First pass of the two-pass 2-D transform: whole row FFTs plus the first radix of the column FFT.
//
// Role in the reference: the per-dimension launches of dispatch_dimensions
// (/root/reference/src/portfft/committed_descriptor_impl.hpp:923-948: one 1-D launch per dimension and per
// (batch, outer index)).  A large 2-D transform (n0 x n1, one matrix far beyond LDS) here takes two launches, and
// the split between them is chosen for HBM access shape instead of "rows, then columns":
//
//   column index r = M*a + b  (a < RC, b < M = n0 / RC),  output row kr = u + RC*v  (u < RC, v < M)
//   W_n0^(r*kr) = W_RC^(a*u) * W_n0^(b*u) * W_M^(b*v)
//
//   pass 1 (this kernel): work-group b of a matrix loads the RC rows {M*a + b} (each a contiguous row), runs the
//     n1-point row FFT on each, then the radix-RC butterfly over a on every column, multiplies by W_n0^(b*u) and
//     stores Y[b][u][kc] as row RC*b + u of the output: RC adjacent rows, one contiguous block.  Both sides are
//     whole rows -- the access shape of the packed 1-D kernels.
//   pass 2 (stockham_strided.hpp, unchanged): the remaining M-point column FFTs over b for every (u, kc).  Rows
//     u + RC*b are RC*n1 elements apart and (u, kc) is one contiguous index of RC*n1 "columns", so this is a
//     batch-interleaved transform of length M over RC*n1 columns: with M = n0/RC short, a work-group holds RC
//     times more adjacent columns than a full-length column pass could (C5: 32-64 columns of 8 B instead of 16).
//
// The radix-RC column butterfly is fused with the last row pass as one 2-D butterfly in registers (R_last x RC
// values per lane), so the data crosses LDS exactly as often as in the packed kernel of length n1.
#pragma once
#include "stockham_wg.hpp"
#include "strided_args.hpp"

namespace pfa {

/// Addressing of pass 1: rows {M*a + b} of one matrix on the input side (lane-contiguous inside a row).
/// SPLIT: separate real / imaginary planes (SPLIT_COMPLEX storage), element size one scalar.
template <typename T, int N, int RC, int AUX, bool SPLIT = false>
struct rows2d_io {
  static constexpr unsigned ES = SPLIT ? sizeof(T) : sizeof(cx<T>);
  __amdgpu_buffer_rsrc_t rin, rout, rin_im, rout_im;
  unsigned row_gap;  // bytes between the rows a and a + 1 of the group (M * N * ES)
  PFA_DEV unsigned in_off(unsigned a, unsigned j) const { return a * row_gap + j * ES; }
  static constexpr unsigned in_step(int k) { return k * ES; }
  // the passes before the last never store; the members exist so that wg_pass<> instantiates
  PFA_DEV unsigned out_off(unsigned a, unsigned j) const { return (a * N + j) * ES; }
  static constexpr unsigned out_step(int k) { return k * ES; }
  PFA_DEV cx<T> load(unsigned voff, unsigned soff) const {
    if constexpr (SPLIT) {
      return {buf_load_scalar<T, AUX>(rin, voff, soff), buf_load_scalar<T, AUX>(rin_im, voff, soff)};
    } else {
      return buf_load<T, AUX>(rin, voff, soff);
    }
  }
  PFA_DEV void store(cx<T> v, unsigned voff, unsigned soff) const {
    if constexpr (SPLIT) {
      buf_store_scalar<T, AUX>(v.re, rout, voff, soff);
      buf_store_scalar<T, AUX>(v.im, rout_im, voff, soff);
    } else {
      buf_store<T, AUX>(v, rout, voff, soff);
    }
  }
};

/// Cfg: wg_cfg of the row FFT with FPW = RC (one LDS image per row of the group).
template <typename Cfg, bool BWD, bool SPLIT = false>
__global__ __launch_bounds__(Cfg::WG, Cfg::OCC) void stockham_rows2d_kernel(const rows2d_args a) {
  using T = typename Cfg::T;
  using Seq = typename Cfg::Seq;
  constexpr int RC = Cfg::FPW;
  constexpr int N = Cfg::N;
  constexpr int NP = Cfg::NP;
  constexpr int PL = NP - 1;           // the last row pass, fused with the column radix
  constexpr int RL = Seq::r[PL];
  constexpr int NBL = N / RL;          // butterflies per row in the last pass; Ns(PL) == NBL
  static_assert(NP >= 2 && !Cfg::STAGED, "rows2d needs a direct-I/O multi-pass row configuration");
  static_assert(NBL % Cfg::WG == 0, "last pass: every lane takes the same butterfly index in all RC rows");
  constexpr int BPTL = NBL / Cfg::WG;
  using IO = rows2d_io<T, N, RC, Cfg::AUX, SPLIT>;
  constexpr unsigned ES = IO::ES;
  extern __shared__ __attribute__((aligned(16))) char pfa_smem[];
  cx<T>* images = reinterpret_cast<cx<T>*>(pfa_smem);
  const int f = threadIdx.x / Cfg::TPF;
  const int tid = threadIdx.x % Cfg::TPF;
  cx<T>* lds = images + f * Cfg::LDS_PER_FFT;
  const cx<T>* __restrict__ tw = static_cast<const cx<T>*>(a.tw);
  const cx<T>* __restrict__ twc = static_cast<const cx<T>*>(a.twc);

  // register-resident twiddles of the middle passes (wg mapping) and of the last pass (flat mapping)
  cx<T> twr[Cfg::TWR_TOTAL];
  [[maybe_unused]] cx<T> twl_r[BPTL][RL - 1];
  if constexpr (Cfg::TWM == TW_REGS) {
    sfor<1, PL>([&](auto p_) PFA_LAMBDA {
      constexpr int p = decltype(p_)::value;
      constexpr int R = Seq::r[p];
      constexpr int Ns = Seq::ns(p);
      sfor<0, Cfg::bpt(p)>([&](auto i_) PFA_LAMBDA {
        constexpr int i = decltype(i_)::value;
        const int q = (tid + i * Cfg::TPF) % Ns;
        sfor<1, R>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          twr[Cfg::twr_off(p) + i * (R - 1) + (t - 1)] = tw[Seq::tw_off(p) + (t - 1) * Ns + q];
        });
      });
    });
    sfor<0, BPTL>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      sfor<1, RL>([&](auto t_) PFA_LAMBDA {
        constexpr int t = decltype(t_)::value;
        twl_r[i][t - 1] = tw[Seq::tw_off(PL) + (t - 1) * NBL + threadIdx.x + i * Cfg::WG];
      });
    });
  }
  if constexpr (Cfg::TWL > 0) {
    cx<T>* twl = images + Cfg::LDS_ELEMS;
    for (int i = threadIdx.x; i < Cfg::TWL_ELEMS; i += Cfg::WG) twl[i] = tw[i];
    __syncthreads();
  }

  const long long M = a.n0 / RC;
  const long long ngroups = a.nmat * M;
  const unsigned row_gap = static_cast<unsigned>(M) * N * ES;
  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const long long m = g / M;
    const unsigned b = static_cast<unsigned>(g - m * M);
    IO io;
    io.row_gap = row_gap;
    {
      const long long ioff = (m * a.n0 + b) * N, ooff = (m * a.n0 + static_cast<long long>(b) * RC) * N;  // elements
      char* ip = const_cast<char*>(static_cast<const char*>(a.in)) + ioff * ES;
      char* op = static_cast<char*>(a.out) + ooff * ES;
      io.rin = __builtin_amdgcn_make_buffer_rsrc(ip, 0, (RC - 1) * row_gap + N * ES, 0x00020000);
      io.rout = __builtin_amdgcn_make_buffer_rsrc(op, 0, RC * N * ES, 0x00020000);
      if constexpr (SPLIT) {
        char* ipi = const_cast<char*>(static_cast<const char*>(a.in_im)) + ioff * ES;
        char* opi = static_cast<char*>(a.out_im) + ooff * ES;
        io.rin_im = __builtin_amdgcn_make_buffer_rsrc(ipi, 0, (RC - 1) * row_gap + N * ES, 0x00020000);
        io.rout_im = __builtin_amdgcn_make_buffer_rsrc(opi, 0, RC * N * ES, 0x00020000);
      } else {
        io.rin_im = io.rin;
        io.rout_im = io.rout;
      }
    }
    const cx<T>* twp = tw;
    if constexpr (Cfg::TWM == TW_GLOBAL) asm volatile("" : "+s"(twp));
    // row passes 0 .. NP-2: exactly the packed kernel's passes (lane = (row, butterfly slot))
    sfor<0, PL>([&](auto p_) PFA_LAMBDA {
      wg_pass<Cfg, BWD, decltype(p_)::value>(io, static_cast<unsigned>(f), lds, tid, twp, twr, T(1));
    });
    // inter-pass column twiddles W_n0^(b*u), uniform over the work-group
    cx<T> colw[RC];
    sfor<1, RC>([&](auto u_) PFA_LAMBDA {
      constexpr int u = decltype(u_)::value;
      colw[u] = twc[(b * static_cast<unsigned>(u)) % static_cast<unsigned>(a.n0)];
    });
    // last row pass x column radix: lane takes butterfly j of all RC rows
    cx<T> v[BPTL][RC][RL];
    sfor<0, BPTL>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      const unsigned j = threadIdx.x + i * Cfg::WG;
      const cx<T>* p = images + lds_pad<Cfg>(j);
      sfor<0, RC>([&](auto r_) PFA_LAMBDA {
        constexpr int r = decltype(r_)::value;
        sfor<0, RL>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          if constexpr (pad_is_linear<Cfg>(NBL, RL, 1)) {
            v[i][r][t] = p[r * Cfg::LDS_PER_FFT + t * pad_step<Cfg>(NBL)];
          } else {
            v[i][r][t] = images[r * Cfg::LDS_PER_FFT + lds_pad<Cfg>(j + t * NBL)];
          }
        });
      });
    });
    __syncthreads();  // the images are free for the next group's pass 0
    sfor<0, BPTL>([&](auto i_) PFA_LAMBDA {
      constexpr int i = decltype(i_)::value;
      const unsigned j = threadIdx.x + i * Cfg::WG;
      cx<T> w[RL];
      sfor<1, RL>([&](auto t_) PFA_LAMBDA {
        constexpr int t = decltype(t_)::value;
        if constexpr (Cfg::TWM == TW_REGS) {
          w[t] = twl_r[i][t - 1];
        } else if constexpr (PL <= Cfg::TWL) {
          w[t] = (images + Cfg::LDS_ELEMS + Seq::tw_off(PL) + (t - 1) * NBL)[j];
        } else {
          w[t] = (twp + Seq::tw_off(PL) + (t - 1) * NBL)[j];
        }
      });
      sfor<0, RC>([&](auto r_) PFA_LAMBDA {
        constexpr int r = decltype(r_)::value;
        sfor<1, RL>([&](auto t_) PFA_LAMBDA {
          constexpr int t = decltype(t_)::value;
          v[i][r][t] = cmul(v[i][r][t], w[t]);
        });
        dft<RL>(v[i][r]);
      });
      sfor<0, RL>([&](auto u_) PFA_LAMBDA {
        constexpr int u = decltype(u_)::value;  // row-FFT output kc = j + u * NBL
        cx<T> c[RC];
        sfor<0, RC>([&](auto r_) PFA_LAMBDA { c[decltype(r_)::value] = v[i][decltype(r_)::value][u]; });
        dft<RC>(c);
        sfor<0, RC>([&](auto k_) PFA_LAMBDA {
          constexpr int k = decltype(k_)::value;  // column-radix output u_col = k -> row RC*b + k
          cx<T> y = c[k];
          if constexpr (k > 0) y = cmul(y, colw[k]);
          if constexpr (BWD) y.im = -y.im;
          io.store(y, j * ES, static_cast<unsigned>((k * N + u * NBL) * ES));
        });
      });
    });
  }
}

}  // namespace pfa
