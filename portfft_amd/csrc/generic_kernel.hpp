// Generic LDS tier: any length that fits LDS, any stride / distance, interleaved or split storage, radices chosen
// at plan time and dispatched at run time.  It is the universal fallback behind the specialised work-group kernels
// and the building block of the strided passes of the N-D and large-N (GLOBAL) plans.
//
// Role in the reference: the UNPACKED / BATCH_INTERLEAVED branches of the work-item, sub-group and work-group
// dispatchers (/root/reference/src/portfft/dispatcher/workitem_dispatcher.hpp:178-204,277-344,
// subgroup_dispatcher.hpp:174-417, workgroup_dispatcher.hpp:148-229) and the store-modifier multiply of the
// global tier (/root/reference/src/portfft/common/global.hpp:135-170).  Design (ours): every FFT of a work-group is
// staged HBM -> LDS with the lane order that makes the HBM side contiguous (element-fastest for packed data,
// batch-fastest for batch-interleaved data), Stockham passes ping-pong between two LDS images, and the result is
// written back with the lane order that suits the output layout.
#pragma once
#include "butterflies.hpp"
#include "generic_args.hpp"

namespace pfa {

template <int R, typename T>
PFA_DEV void generic_butterfly(const cx<T>* __restrict__ a, cx<T>* __restrict__ b, int j, int nb, int ns,
                               const cx<T>* __restrict__ tw) {
  cx<T> v[R];
  sfor<0, R>([&](auto t_) PFA_LAMBDA {
    constexpr int t = decltype(t_)::value;
    v[t] = a[j + t * nb];
  });
  const int q = j % ns;
  if (ns > 1) {
    sfor<1, R>([&](auto t_) PFA_LAMBDA {
      constexpr int t = decltype(t_)::value;
      v[t] = cmul(v[t], tw[(t - 1) * ns + q]);
    });
  }
  dft<R>(v);
  const int base = (j / ns) * (ns * R) + q;
  sfor<0, R>([&](auto u_) PFA_LAMBDA {
    constexpr int u = decltype(u_)::value;
    b[base + u * ns] = v[u];
  });
}

template <typename T>
__global__ __launch_bounds__(GENERIC_WG) void generic_fft_kernel(const generic_args p) {
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_generic[];
  cx<T>* A = reinterpret_cast<cx<T>*>(pfa_smem_generic);
  cx<T>* B = A + static_cast<size_t>(p.fpw) * p.n;
  const int tid = threadIdx.x;
  const int n = p.n;
  const int fpw = p.fpw;
  const T* __restrict__ in_re = static_cast<const T*>(p.in_re);
  const T* __restrict__ in_im = static_cast<const T*>(p.in_im);
  T* __restrict__ out_re = static_cast<T*>(p.out_re);
  T* __restrict__ out_im = static_cast<T*>(p.out_im);
  const cx<T>* __restrict__ tw = static_cast<const cx<T>*>(p.tw);
  const long long ngroups = (p.total_count + fpw - 1) / fpw;
  const T scale = static_cast<T>(p.scale);

  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    // ---- stage in ----
    for (int e = tid; e < fpw * n; e += GENERIC_WG) {
      const int f = p.in_f_fast ? e % fpw : e / n;
      const int i = p.in_f_fast ? e / fpw : e % n;
      const long long t = g * fpw + f;
      if (t < p.total_count) {
        const long long idx = (t / p.inner_count) * p.in_dist_outer + (t % p.inner_count) * p.in_dist_inner +
                              static_cast<long long>(i) * p.in_stride;
        cx<T> x = {in_re[idx * p.in_step], in_im[idx * p.in_step]};
        if (p.conj_in) x.im = -x.im;
        A[f * n + i] = x;
      }
    }
    __syncthreads();
    // ---- Stockham passes, LDS image A -> B ----
    cx<T>* src = A;
    cx<T>* dst = B;
    int ns = 1;
    for (int pass = 0; pass < p.n_passes; ++pass) {
      const int R = p.radix[pass];
      const int nb = n / R;
      const cx<T>* twp = tw + p.tw_off[pass];
      for (int w = tid; w < fpw * nb; w += GENERIC_WG) {
        const int f = w / nb;
        const int j = w % nb;
        const cx<T>* a = src + f * n;
        cx<T>* b = dst + f * n;
        switch (R) {
#define PFA_CASE(r)                            \
  case r:                                      \
    generic_butterfly<r>(a, b, j, nb, ns, twp); \
    break;
          PFA_GENERIC_RADICES(PFA_CASE)
#undef PFA_CASE
          default:
            break;
        }
      }
      __syncthreads();
      cx<T>* tmp = src;
      src = dst;
      dst = tmp;
      ns *= R;
    }
    // ---- stage out ----
    for (int e = tid; e < fpw * n; e += GENERIC_WG) {
      const int f = p.out_f_fast ? e % fpw : e / n;
      const int k = p.out_f_fast ? e / fpw : e % n;
      const long long t = g * fpw + f;
      if (t < p.total_count) {
        cx<T> y = src[f * n + k];
        const long long c = t % p.inner_count;
        if (p.stw_lo != nullptr) {
          const long long m = static_cast<long long>(k) * c;
          const cx<T> wl = static_cast<const cx<T>*>(p.stw_lo)[m & ((1ll << p.stw_shift) - 1)];
          const cx<T> wh = static_cast<const cx<T>*>(p.stw_hi)[m >> p.stw_shift];
          y = cmul(y, cmul(wl, wh));
        }
        if (p.conj_out) y.im = -y.im;
        y.re *= scale;
        y.im *= scale;
        const long long idx = (t / p.inner_count) * p.out_dist_outer + c * p.out_dist_inner +
                              static_cast<long long>(k) * p.out_stride;
        out_re[idx * p.out_step] = y.re;
        out_im[idx * p.out_step] = y.im;
      }
    }
    __syncthreads();
  }
}

}  // namespace pfa
