// In-register DFT butterflies for CDNA4 lanes.
//
// Role in the reference: wi_dft / cooley_tukey_dft / naive_dft
// (/root/reference/src/portfft/common/workitem.hpp:64-127,200-219) -- the DFT a single work-item performs on
// values held in its private registers.  This is an independent design: radices are template constants, every
// root of unity is an immediate (radix_constants.inc), composite radices are split 4xB (or smallest-prime x B)
// fully unrolled, and odd primes use the symmetric half-length form (P-1)/2 cos/sin sums instead of the
// reference's O(P^2) complex multiply loop.
//
// Sign convention: forward transform, W_R = exp(-2*pi*i/R).  The backward transform is obtained by the callers
// through conjugation on load and store (same identity the reference uses, committed_descriptor_impl.hpp:469-472).
#pragma once
// Under hiprtc (runtime specialisation, jit.cpp) the HIP runtime declarations are pre-included and no standard
// library is available, so these headers use none.
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

namespace pfa {

/// compile-time integer tag (what sfor hands to its body)
template <int V>
struct int_tag {
  static constexpr int value = V;
};
/// unevaluated-operand helper
template <typename T>
T&& declval_of() noexcept;

#define PFA_DEV __host__ __device__ __forceinline__
#define PFA_LAMBDA __attribute__((always_inline))

template <typename T>
struct alignas(2 * sizeof(T)) cx {
  T re, im;
};

template <typename T>
PFA_DEV cx<T> operator+(cx<T> a, cx<T> b) {
  return {a.re + b.re, a.im + b.im};
}
template <typename T>
PFA_DEV cx<T> operator-(cx<T> a, cx<T> b) {
  return {a.re - b.re, a.im - b.im};
}
template <typename T>
PFA_DEV cx<T> cmul(cx<T> a, cx<T> b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <typename T>
PFA_DEV cx<T> cconj(cx<T> a) {
  return {a.re, -a.im};
}
/// multiply by -i
template <typename T>
PFA_DEV cx<T> mul_mi(cx<T> a) {
  return {a.im, -a.re};
}
/// multiply by +i
template <typename T>
PFA_DEV cx<T> mul_pi(cx<T> a) {
  return {-a.im, a.re};
}

/// a + (-i)*b and a - (-i)*b
template <typename T>
PFA_DEV cx<T> add_mi(cx<T> a, cx<T> b) {
  return {a.re + b.im, a.im - b.re};
}
template <typename T>
PFA_DEV cx<T> sub_mi(cx<T> a, cx<T> b) {
  return {a.re - b.im, a.im + b.re};
}
/// a * (c - i*s) for a constant pair (c, s)
template <typename T>
PFA_DEV cx<T> mul_cs(cx<T> a, T c, T s) {
  return {a.re * c + a.im * s, a.im * c - a.re * s};
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(PFA_NO_PK_ASM)
// fp32 on gfx950: one complex value = one 64-bit VGPR pair, arithmetic = packed-f32 VOP3P.  hipcc (ROCm 7.2) does not
// use the per-half op_sel / neg modifiers of v_pk_*_f32: a complex multiply becomes 3 packed ops + a move, a
// "+/- i" rotation 2 packed ops + a move (37 % of the N=4096 kernel's VALU instructions were moves).  The forms
// below are the 2- and 1-instruction sequences; each asm statement is a single VALU instruction, so the scheduler
// stays free to interleave them.
typedef float pfa_v2f __attribute__((ext_vector_type(2)));
PFA_DEV pfa_v2f to_v2(cx<float> a) { return __builtin_bit_cast(pfa_v2f, a); }
PFA_DEV cx<float> from_v2(pfa_v2f a) { return __builtin_bit_cast(cx<float>, a); }

template <>
PFA_DEV cx<float> cmul<float>(cx<float> a, cx<float> b) {
  pfa_v2f t, r;
  const pfa_v2f x = to_v2(a), y = to_v2(b);
  // t = (a.re*b.re, a.re*b.im);  r = (-a.im*b.im + t.lo, a.im*b.re + t.hi)
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "v"(y));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(x), "v"(y), "v"(t));
  return from_v2(r);
}
template <>
PFA_DEV cx<float> add_mi<float>(cx<float> a, cx<float> b) {
  pfa_v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(to_v2(a)), "v"(to_v2(b)));
  return from_v2(r);
}
template <>
PFA_DEV cx<float> sub_mi<float>(cx<float> a, cx<float> b) {
  pfa_v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(to_v2(a)), "v"(to_v2(b)));
  return from_v2(r);
}
template <>
PFA_DEV cx<float> mul_cs<float>(cx<float> a, float c, float s) {
  pfa_v2f t, r;
  const pfa_v2f x = to_v2(a);
  const pfa_v2f k = {c, s};
  // t = (a.re*c, a.im*c);  r = (a.im*s + t.lo, -a.re*s + t.hi)
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "s"(k));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(x), "s"(k), "v"(t));
  return from_v2(r);
}
#endif

#include "radix_constants.inc"

/// compile-time loop: f(int_tag<I>) for I in [B, E)
template <int B, int E, typename F>
PFA_DEV void sfor(F&& f) {
  if constexpr (B < E) {
    f(int_tag<B>{});
    sfor<B + 1, E>(f);
  }
}

/// a * W_R^K with W_R = exp(-2*pi*i/R); K and R are compile-time so trivial roots cost nothing.
template <int R, int K, typename T>
PFA_DEV cx<T> mul_root(cx<T> a) {
  constexpr int k = ((K % R) + R) % R;
  if constexpr (k == 0) {
    return a;
  } else if constexpr (4 * k == R) {
    return mul_mi(a);
  } else if constexpr (2 * k == R) {
    return {-a.re, -a.im};
  } else if constexpr (4 * k == 3 * R) {
    return mul_pi(a);
  } else {
    constexpr T c = static_cast<T>(unit_roots<R>::c[k]);
    constexpr T s = static_cast<T>(unit_roots<R>::s[k]);
    return mul_cs<T>(a, c, s);
  }
}

constexpr bool is_prime(int n) {
  if (n < 2) return false;
  for (int i = 2; i * i <= n; ++i) {
    if (n % i == 0) return false;
  }
  return true;
}

/// first factor of the in-register Cooley-Tukey split: 4 when possible, otherwise the smallest prime factor.
constexpr int ct_split(int r) {
  if (r % 4 == 0) return 4;
  for (int i = 2; i * i <= r; ++i) {
    if (r % i == 0) return i;
  }
  return r;
}

template <int R, typename T>
PFA_DEV void dft(cx<T> (&v)[R]);

/// odd prime length: pair v[k], v[P-k]; X[u], X[P-u] = m -/+ i*n with real-coefficient sums m, n.
template <int P, typename T>
PFA_DEV void dft_odd_prime(cx<T> (&v)[P]) {
  constexpr int H = (P - 1) / 2;
  cx<T> tp[H], tm[H];
  sfor<0, H>([&](auto k_) PFA_LAMBDA {
    constexpr int k = decltype(k_)::value;
    tp[k] = v[k + 1] + v[P - 1 - k];
    tm[k] = v[k + 1] - v[P - 1 - k];
  });
  const cx<T> x0 = v[0];
  cx<T> sum = x0;
  sfor<0, H>([&](auto k_) PFA_LAMBDA { sum = sum + tp[decltype(k_)::value]; });
  v[0] = sum;
  sfor<1, H + 1>([&](auto u_) PFA_LAMBDA {
    constexpr int u = decltype(u_)::value;
    cx<T> m = x0;
    cx<T> n = {T(0), T(0)};
    sfor<1, H + 1>([&](auto k_) PFA_LAMBDA {
      constexpr int k = decltype(k_)::value;
      constexpr int idx = (k * u) % P;
      constexpr T c = static_cast<T>(unit_roots<P>::c[idx]);
      constexpr T s = static_cast<T>(unit_roots<P>::s[idx]);
      m.re += c * tp[k - 1].re;
      m.im += c * tp[k - 1].im;
      n.re += s * tm[k - 1].re;
      n.im += s * tm[k - 1].im;
    });
    v[u] = {m.re + n.im, m.im - n.re};
    v[P - u] = {m.re - n.im, m.im + n.re};
  });
}

/// In-place forward DFT of R values held in registers; natural order in and out.
template <int R, typename T>
PFA_DEV void dft(cx<T> (&v)[R]) {
  if constexpr (R == 1) {
  } else if constexpr (R == 2) {
    const cx<T> a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
  } else if constexpr (R == 4) {
    const cx<T> a = v[0] + v[2], b = v[0] - v[2];
    const cx<T> c = v[1] + v[3], e = v[1] - v[3];
    v[0] = a + c;
    v[1] = add_mi(b, e);
    v[2] = a - c;
    v[3] = sub_mi(b, e);
  } else if constexpr (is_prime(R)) {
    dft_odd_prime<R>(v);
  } else {
    constexpr int A = ct_split(R);
    constexpr int B = R / A;
    cx<T> u[B][A];
    sfor<0, B>([&](auto b_) PFA_LAMBDA {
      constexpr int b = decltype(b_)::value;
      sfor<0, A>([&](auto a_) PFA_LAMBDA {
        constexpr int a = decltype(a_)::value;
        u[b][a] = v[a * B + b];
      });
      dft<A>(u[b]);
      sfor<1, A>([&](auto k_) PFA_LAMBDA {
        constexpr int k1 = decltype(k_)::value;
        u[b][k1] = mul_root<R, b * k1>(u[b][k1]);
      });
    });
    sfor<0, A>([&](auto k_) PFA_LAMBDA {
      constexpr int k1 = decltype(k_)::value;
      cx<T> w[B];
      sfor<0, B>([&](auto b_) PFA_LAMBDA { w[decltype(b_)::value] = u[decltype(b_)::value][k1]; });
      dft<B>(w);
      sfor<0, B>([&](auto k2_) PFA_LAMBDA { v[k1 + A * decltype(k2_)::value] = w[decltype(k2_)::value]; });
    });
  }
}

}  // namespace pfa
