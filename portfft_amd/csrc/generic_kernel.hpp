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

PFA_DEV unsigned fast_div(unsigned e, unsigned long long magic) {
  return static_cast<unsigned>((static_cast<unsigned long long>(e) * magic) >> 40);
}

template <int R, typename T>
PFA_DEV void generic_butterfly(const cx<T>* __restrict__ a, cx<T>* __restrict__ b, unsigned j, unsigned nb,
                               unsigned ns, unsigned long long magic_ns, const cx<T>* __restrict__ tw) {
  cx<T> v[R];
  sfor<0, R>([&](auto t_) PFA_LAMBDA {
    constexpr int t = decltype(t_)::value;
    v[t] = a[j + t * nb];
  });
  const unsigned jq = fast_div(j, magic_ns);
  const unsigned q = j - jq * ns;
  if (ns > 1) {
    sfor<1, R>([&](auto t_) PFA_LAMBDA {
      constexpr int t = decltype(t_)::value;
      v[t] = cmul(v[t], tw[(t - 1) * ns + q]);
    });
  }
  dft<R>(v);
  const unsigned base = jq * (ns * R) + q;
  sfor<0, R>([&](auto u_) PFA_LAMBDA {
    constexpr int u = decltype(u_)::value;
    b[base + u * ns] = v[u];
  });
}

/// BIG: the instantiation that also carries the prime radices 37 ... 61 (generic_args.hpp)
template <typename T, bool BIG = false>
__global__ __launch_bounds__(GENERIC_WG) void generic_fft_kernel(const generic_args p) {
  extern __shared__ __attribute__((aligned(16))) char pfa_smem_generic[];
  cx<T>* A = reinterpret_cast<cx<T>*>(pfa_smem_generic);
  cx<T>* B = A + static_cast<size_t>(p.fpw) * p.n;
  const unsigned tid = threadIdx.x;
  const unsigned n = p.n;
  const unsigned fpw = p.fpw;
  const T* in_re = static_cast<const T*>(p.in_re);
  const T* in_im = static_cast<const T*>(p.in_im);
  T* out_re = static_cast<T*>(p.out_re);
  T* out_im = static_cast<T*>(p.out_im);
  const cx<T>* __restrict__ tw = static_cast<const cx<T>*>(p.tw);
  const long long ngroups = (p.total_count + fpw - 1) / fpw;
  const T scale = static_cast<T>(p.scale);

  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    // position of the group's first FFT: one 64-bit division per group, none per element
    const long long t0 = g * fpw;
    const long long o0 = t0 / p.inner_count;
    const long long c0 = t0 - o0 * p.inner_count;
    const long long left = p.total_count - t0;
    const unsigned nf = left < static_cast<long long>(fpw) ? static_cast<unsigned>(left) : fpw;
    // ---- stage in ----
    for (unsigned e = tid; e < fpw * n; e += GENERIC_WG) {
      unsigned f, i;
      if (p.in_f_fast) {
        i = fast_div(e, p.magic_fpw);
        f = e - i * fpw;
      } else {
        f = fast_div(e, p.magic_n);
        i = e - f * n;
      }
      if (f < nf) {
        long long c = c0 + f, o = o0;
        while (c >= p.inner_count) {
          c -= p.inner_count;
          ++o;
        }
        const long long idx = o * p.in_dist_outer + c * p.in_dist_inner + static_cast<long long>(i) * p.in_stride;
        cx<T> x = {in_re[idx * p.in_step], in_im[idx * p.in_step]};
        if (p.conj_in) x.im = -x.im;
        A[f * n + i] = x;
      }
    }
    __syncthreads();
    // ---- Stockham passes, LDS image A -> B ----
    cx<T>* src = A;
    cx<T>* dst = B;
    unsigned ns = 1;
    for (int pass = 0; pass < p.n_passes; ++pass) {
      const unsigned R = p.radix[pass];
      const unsigned nb = n / R;
      const unsigned long long mnb = p.magic_nb[pass];
      const unsigned long long mns = p.magic_ns[pass];
      const cx<T>* twp = tw + p.tw_off[pass];
      for (unsigned w = tid; w < nf * nb; w += GENERIC_WG) {
        const unsigned f = fast_div(w, mnb);
        const unsigned j = w - f * nb;
        const cx<T>* a = src + f * n;
        cx<T>* b = dst + f * n;
        switch (R) {
#define PFA_CASE(r)                                      \
  case r:                                                \
    generic_butterfly<r>(a, b, j, nb, ns, mns, twp); \
    break;
          PFA_GENERIC_RADICES(PFA_CASE)
#undef PFA_CASE
          default:
            if constexpr (BIG) {
              switch (R) {
#define PFA_CASE(r)                                      \
  case r:                                                \
    generic_butterfly<r>(a, b, j, nb, ns, mns, twp); \
    break;
                PFA_GENERIC_RADICES_BIG(PFA_CASE)
#undef PFA_CASE
                default:
                  break;
              }
            }
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
    for (unsigned e = tid; e < fpw * n; e += GENERIC_WG) {
      unsigned f, k;
      if (p.out_f_fast) {
        k = fast_div(e, p.magic_fpw);
        f = e - k * fpw;
      } else {
        f = fast_div(e, p.magic_n);
        k = e - f * n;
      }
      if (f < nf) {
        long long c = c0 + f, o = o0;
        while (c >= p.inner_count) {
          c -= p.inner_count;
          ++o;
        }
        cx<T> y = src[f * n + k];
        if (p.stw_lo != nullptr) {
          const long long m = static_cast<long long>(k) * c;
          const cx<T> wl = static_cast<const cx<T>*>(p.stw_lo)[m & ((1ll << p.stw_shift) - 1)];
          const cx<T> wh = static_cast<const cx<T>*>(p.stw_hi)[m >> p.stw_shift];
          y = cmul(y, cmul(wl, wh));
        }
        if (p.conj_out) y.im = -y.im;
        y.re *= scale;
        y.im *= scale;
        const long long idx = o * p.out_dist_outer + c * p.out_dist_inner + static_cast<long long>(k) * p.out_stride;
        out_re[idx * p.out_step] = y.re;
        out_im[idx * p.out_step] = y.im;
      }
    }
    __syncthreads();
  }
}

}  // namespace pfa
