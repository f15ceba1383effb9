/*
 * pfft_oracle.c -- CPU restatement of portFFT's C2C execute path (TEST INFRASTRUCTURE ONLY, see pfft_oracle.h).
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC -> oracle/libpfft_oracle.so)
 */
#define _GNU_SOURCE
#include "pfft_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* CMake defaults of the reference (CMakeLists.txt:53-94, src/portfft/defines.hpp:33-35) */
#define PORTFFT_REGISTERS_PER_WI 128
#define PORTFFT_SGS_IN_WG 2
#define PORTFFT_N_LOCAL_BANKS 32

static void set_msg(char* msg, size_t msglen, const char* fmt, ...) {
  if (msg == NULL || msglen == 0) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, msglen, fmt, ap);
  va_end(ap);
}

int32_t pfo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---------------------------------------------------------------------------------------------------------------
 * planner predicates
 * ------------------------------------------------------------------------------------------------------------- */

/* common/workitem.hpp:135-144: largest divisor i with i*i <= N (1 if prime) */
int64_t pfo_factorize(int64_t n) {
  int64_t res = 1;
  for (int64_t i = 2; i * i <= n; i++) {
    if (n % i == 0) res = i;
  }
  return res;
}

/* common/workitem.hpp:154-169, MaxRecursionLevel = int_log2(56) - 1 = 4 */
static int64_t wi_temps_rec(int64_t n, int level) {
  int64_t f0 = pfo_factorize(n);
  int64_t f1 = n / f0;
  if (f0 < 2 || f1 < 2) return n;
  int64_t a = 2, b = 2;
  if (level < 4) {
    a = wi_temps_rec(f0, level + 1);
    b = wi_temps_rec(f1, level + 1);
  }
  return (a > b ? a : b) + n;
}
int64_t pfo_wi_temps(int64_t n) { return wi_temps_rec(n, 0); }

/* common/workitem.hpp:179-185 */
int32_t pfo_fits_in_wi(int64_t n, int32_t scalar_bytes) {
  int64_t n_complex = n + pfo_wi_temps(n);
  int64_t complex_size = 2 * (int64_t)scalar_bytes;
  int64_t register_space = PORTFFT_REGISTERS_PER_WI * 4;
  return n_complex * complex_size <= register_space;
}

/* common/subgroup.hpp:226-238 */
int64_t pfo_factorize_sg(int64_t n, int32_t sg_size) {
  for (int64_t i = sg_size; i > 1; i--) {
    if (n % i == 0) return i;
  }
  return 1;
}

/* common/subgroup.hpp:248-253 */
int32_t pfo_fits_in_sg(int64_t n, int32_t scalar_bytes, int32_t sg_size) {
  int64_t factor_sg = pfo_factorize_sg(n, sg_size);
  int64_t factor_wi = n / factor_sg;
  return pfo_fits_in_wi(factor_wi, scalar_bytes);
}

/* common/memory_views.hpp:94-101 (pad_local) and common/workgroup.hpp:44-53 (bank_lines_per_pad_wg) */
static int64_t pad_local(int64_t local_idx, int64_t bank_lines_per_pad) {
  return local_idx + local_idx / (PORTFFT_N_LOCAL_BANKS * bank_lines_per_pad);
}
static int64_t bank_lines_per_pad_wg(int64_t row_size) {
  const int64_t bank_line_size = (int64_t)sizeof(float) * PORTFFT_N_LOCAL_BANKS;
  if (row_size % bank_line_size == 0) return row_size / bank_line_size;
  return 1;
}

/* dispatcher/workgroup_dispatcher.hpp:364-380 (PACKED: one batch in local memory) */
static int64_t num_scalars_in_local_mem_workgroup(int64_t length, int64_t n, int64_t m, int32_t scalar_bytes,
                                                  int batch_interleaved, int32_t sg_size) {
  int64_t num_batches = batch_interleaved ? (int64_t)sg_size * PORTFFT_SGS_IN_WG / 2 : 1;
  return pad_local(2 * num_batches * length, bank_lines_per_pad_wg(2 * (int64_t)scalar_bytes * m)) + 2 * (m + n);
}

/* dispatcher/subgroup_dispatcher.hpp:774-803 */
static int64_t num_scalars_in_local_mem_subgroup(int64_t length, int64_t factor_sg, int32_t scalar_bytes,
                                                 int32_t sg_size, int64_t local_mem_bytes, int batch_interleaved) {
  int64_t dft_length = length;
  int64_t twiddle_bytes = 2 * dft_length * scalar_bytes;
  if (batch_interleaved) {
    int64_t padded_fft_bytes = pad_local(2 * dft_length, 1) * scalar_bytes;
    int64_t max_batches = (local_mem_bytes - twiddle_bytes) / padded_fft_bytes;
    int64_t batches_per_sg = sg_size / 2;
    int64_t q = max_batches / batches_per_sg;
    int64_t num_sgs = q < 1 ? 1 : q;
    if (num_sgs > PORTFFT_SGS_IN_WG) num_sgs = PORTFFT_SGS_IN_WG;
    int64_t num_batches = sg_size * num_sgs / 2;
    return pad_local(2 * dft_length * num_batches, 1);
  }
  int64_t n_ffts_per_sg = sg_size / factor_sg;
  int64_t num_scalars_per_sg = pad_local(2 * dft_length * n_ffts_per_sg, 1);
  int64_t max_n_sgs = (local_mem_bytes - twiddle_bytes) / scalar_bytes / num_scalars_per_sg;
  int64_t num_sgs = max_n_sgs < 1 ? 1 : max_n_sgs;
  if (num_sgs > PORTFFT_SGS_IN_WG) num_sgs = PORTFFT_SGS_IN_WG;
  return pad_local(2 * dft_length * n_ffts_per_sg * num_sgs, 1);
}

/* the lambda check_and_select_target_level of committed_descriptor_impl.hpp:269-309 */
static int check_and_select_target_level(int64_t factor_size, int batch_interleaved, int32_t scalar_bytes,
                                         int32_t sg_size, int64_t local_mem_bytes, pfo_impl_t* impl) {
  if (pfo_fits_in_wi(factor_size, scalar_bytes)) {
    int k = impl->n_kernels++;
    impl->kernel_level[k] = PFO_WORKITEM;
    impl->kernel_length[k] = factor_size;
    impl->n_factors[k] = 1;
    impl->factors[k][0] = (int32_t)factor_size;
    return 1;
  }
  int64_t factor_sg = pfo_factorize_sg(factor_size, sg_size);
  int64_t factor_wi = factor_size / factor_sg;
  int64_t input_scalars = num_scalars_in_local_mem_subgroup(factor_size, factor_sg, scalar_bytes, sg_size,
                                                            local_mem_bytes, batch_interleaved);
  int64_t store_modifiers = batch_interleaved ? input_scalars : 0;
  int64_t twiddle_scalars = 2 * factor_size;
  int fits_local = (int64_t)scalar_bytes * (input_scalars + store_modifiers + twiddle_scalars) < local_mem_bytes;
  if (pfo_fits_in_sg(factor_size, scalar_bytes, sg_size) && fits_local) {
    int k = impl->n_kernels++;
    impl->kernel_level[k] = PFO_SUBGROUP;
    impl->kernel_length[k] = factor_size;
    impl->n_factors[k] = 2;
    impl->factors[k][0] = (int32_t)factor_sg; /* order used inside GLOBAL: {factor_sg, factor_wi} (:301) */
    impl->factors[k][1] = (int32_t)factor_wi;
    return 1;
  }
  return 0;
}

/* utils.hpp:94-113 (factorize_input_impl): returns 0 on success, status otherwise */
static int32_t factorize_input_impl(int64_t factor_size, int transposed, int32_t scalar_bytes, int32_t sg_size,
                                    int64_t local_mem_bytes, pfo_impl_t* impl, int64_t* out_factor, char* msg,
                                    size_t msglen) {
  int64_t fact_1 = factor_size;
  if (check_and_select_target_level(fact_1, transposed, scalar_bytes, sg_size, local_mem_bytes, impl)) {
    *out_factor = fact_1;
    return PFFT_OK;
  }
  if (pfo_factorize(fact_1) == 1) {
    set_msg(msg, msglen, "Large prime sized factors are not supported at the moment");
    return PFFT_UNSUPPORTED_CONFIGURATION;
  }
  do {
    fact_1 = pfo_factorize(fact_1);
    if (fact_1 == 1) {
      set_msg(msg, msglen, "Factorization Failed !");
      return PFFT_INTERNAL_ERROR;
    }
    /* the lambda's second parameter defaults to batch_interleaved_layout = true (:269) */
  } while (!check_and_select_target_level(fact_1, 1, scalar_bytes, sg_size, local_mem_bytes, impl));
  *out_factor = fact_1;
  return PFFT_OK;
}

/* committed_descriptor_impl.hpp:210-313 */
int32_t pfo_prepare_implementation(int64_t fft_size, int32_t scalar_bytes, int32_t sg_size, int64_t local_mem_bytes,
                                   pfo_impl_t* impl, char* msg, size_t msglen) {
  memset(impl, 0, sizeof(*impl));
  if (pfo_fits_in_wi(fft_size, scalar_bytes)) {
    impl->level = PFO_WORKITEM;
    impl->n_kernels = 1;
    impl->kernel_level[0] = PFO_WORKITEM;
    impl->kernel_length[0] = fft_size;
    return PFFT_OK;
  }
  if (pfo_fits_in_sg(fft_size, scalar_bytes, sg_size)) {
    int64_t factor_sg = pfo_factorize_sg(fft_size, sg_size);
    int64_t factor_wi = fft_size / factor_sg;
    impl->level = PFO_SUBGROUP;
    impl->n_kernels = 1;
    impl->kernel_level[0] = PFO_SUBGROUP;
    impl->kernel_length[0] = fft_size;
    impl->n_factors[0] = 2;
    impl->factors[0][0] = (int32_t)factor_wi;
    impl->factors[0][1] = (int32_t)factor_sg;
    return PFFT_OK;
  }
  int64_t n_idx = pfo_factorize(fft_size);
  if (n_idx <= INT32_MAX && fft_size / n_idx <= INT32_MAX) {
    if (n_idx == 1) {
      set_msg(msg, msglen, "FFT size %lld : Large Prime sized FFT currently is unsupported", (long long)fft_size);
      return PFFT_UNSUPPORTED_CONFIGURATION;
    }
    int64_t n = n_idx, m = fft_size / n_idx;
    int64_t factor_sg_n = pfo_factorize_sg(n, sg_size), factor_wi_n = n / factor_sg_n;
    int64_t factor_sg_m = pfo_factorize_sg(m, sg_size), factor_wi_m = m / factor_sg_m;
    int64_t local_memory_usage =
        num_scalars_in_local_mem_workgroup(fft_size, n, m, scalar_bytes, 0, sg_size) * scalar_bytes;
    if (pfo_fits_in_wi(factor_wi_n, scalar_bytes) && pfo_fits_in_wi(factor_wi_m, scalar_bytes) &&
        local_memory_usage <= local_mem_bytes) {
      impl->level = PFO_WORKGROUP;
      impl->n_kernels = 1;
      impl->kernel_level[0] = PFO_WORKGROUP;
      impl->kernel_length[0] = fft_size;
      impl->n_factors[0] = 4;
      impl->factors[0][0] = (int32_t)factor_wi_n;
      impl->factors[0][1] = (int32_t)factor_sg_n;
      impl->factors[0][2] = (int32_t)factor_wi_m;
      impl->factors[0][3] = (int32_t)factor_sg_m;
      return PFFT_OK;
    }
  }
  /* utils.hpp:122-132 (factorize_input) */
  impl->level = PFO_GLOBAL;
  impl->n_kernels = 0;
  if (pfo_factorize(fft_size) == 1) {
    set_msg(msg, msglen, "Large Prime sized FFTs are currently not supported");
    return PFFT_UNSUPPORTED_CONFIGURATION;
  }
  int64_t temp = 1;
  while (fft_size / temp != 1) {
    int64_t f = 0;
    if (impl->n_kernels >= PFO_MAX_FACTORS - 1) {
      set_msg(msg, msglen, "too many factors");
      return PFFT_INTERNAL_ERROR;
    }
    int32_t st = factorize_input_impl(fft_size / temp, 1, scalar_bytes, sg_size, local_mem_bytes, impl, &f, msg, msglen);
    if (st != PFFT_OK) return st;
    temp *= f;
  }
  return PFFT_OK;
}

/* ---------------------------------------------------------------------------------------------------------------
 * static twiddle table
 * ------------------------------------------------------------------------------------------------------------- */

/* scripts/generate_twiddles.py:60-92: exact values on the axes, otherwise libm cos/sin of -2*pi*i/size in double */
static double static_twiddle_compute(int32_t n, int32_t k, int32_t imag);
/* the reference holds the values in a 65 x 65 constexpr table (common/twiddle.hpp:25-155); cache them the same way */
static double g_tw_table[2][65][65];
static int g_tw_ready = 0;
static void static_twiddle_init(void) {
  if (g_tw_ready) return;
#pragma omp critical(pfo_tw_init)
  {
    if (!g_tw_ready) {
      for (int n = 0; n < 65; ++n)
        for (int k = 0; k < 65; ++k) {
          g_tw_table[0][n][k] = static_twiddle_compute(n, k, 0);
          g_tw_table[1][n][k] = static_twiddle_compute(n, k, 1);
        }
      g_tw_ready = 1;
    }
  }
}
double pfo_static_twiddle(int32_t n, int32_t k, int32_t imag) {
  if (n < 0 || n > 64 || k < 0 || k > 64) return static_twiddle_compute(n, k, imag);
  if (!g_tw_ready) static_twiddle_init();
  return g_tw_table[imag ? 1 : 0][n][k];
}
static double static_twiddle_compute(int32_t n, int32_t k, int32_t imag) {
  if (n <= 0 || k < 0 || k >= n) return 0.0; /* zero padding of the table */
  if (k == 0) return imag ? 0.0 : 1.0;
  if (2 * k == n) return imag ? 0.0 : -1.0;
  if (4 * k == n) return imag ? -1.0 : 0.0;
  if (4 * k == n * 3) return imag ? 1.0 : 0.0;
  double theta = -2. * M_PI * k / n;
  return imag ? sin(theta) : cos(theta);
}

/* ---------------------------------------------------------------------------------------------------------------
 * descriptor logic
 * ------------------------------------------------------------------------------------------------------------- */

static const uint64_t* strides_of(const pfft_desc_t* d, int dir) {
  return dir == PFFT_FORWARD ? d->forward_strides : d->backward_strides;
}
static int n_strides_of(const pfft_desc_t* d, int dir) {
  return dir == PFFT_FORWARD ? d->n_forward_strides : d->n_backward_strides;
}
static uint64_t distance_of(const pfft_desc_t* d, int dir) {
  return dir == PFFT_FORWARD ? d->forward_distance : d->backward_distance;
}
static uint64_t offset_of(const pfft_desc_t* d, int dir) {
  return dir == PFFT_FORWARD ? d->forward_offset : d->backward_offset;
}

uint64_t pfo_flattened_length(const pfft_desc_t* d) {
  uint64_t t = 1;
  for (int i = 0; i < d->rank; ++i) t *= d->lengths[i];
  return t;
}

/* descriptor.hpp:262-270 */
uint64_t pfo_input_count(const pfft_desc_t* d, int32_t dir) {
  const uint64_t* strides = strides_of(d, dir);
  uint64_t last = (d->number_of_transforms - 1) * distance_of(d, dir);
  for (int i = 0; i < d->rank; ++i) last += (d->lengths[i] - 1) * strides[i];
  return offset_of(d, dir) + last + 1;
}
uint64_t pfo_output_count(const pfft_desc_t* d, int32_t dir) {
  return pfo_input_count(d, dir == PFFT_FORWARD ? PFFT_BACKWARD : PFFT_FORWARD);
}

/* utils.hpp:190-246 */
int32_t pfo_layout(const pfft_desc_t* d, int32_t dir) {
  const uint64_t* strides = strides_of(d, dir);
  int is_default = n_strides_of(d, dir) == d->rank && distance_of(d, dir) == pfo_flattened_length(d);
  uint64_t total = 1;
  for (int i = d->rank - 1; i >= 0 && is_default; --i) {
    if (strides[i] != total) is_default = 0;
    total *= d->lengths[i];
  }
  if (is_default) return PFFT_LAYOUT_PACKED;
  if (d->rank == 1 && distance_of(d, dir) == 1 && strides[d->rank - 1] == d->number_of_transforms)
    return PFFT_LAYOUT_BATCH_INTERLEAVED;
  return PFFT_LAYOUT_UNPACKED;
}

/* descriptor_validation.hpp:92-111 */
static int32_t validate_strides_distance_basic(const pfft_desc_t* d, int dir, const char* name, char* msg,
                                               size_t msglen) {
  const uint64_t* strides = strides_of(d, dir);
  if (n_strides_of(d, dir) != d->rank) {
    set_msg(msg, msglen, "Mismatching %s strides length got %d expected %d", name, n_strides_of(d, dir), d->rank);
    return PFFT_INVALID_CONFIGURATION;
  }
  for (int i = 0; i < d->rank; ++i) {
    if (strides[i] == 0) {
      set_msg(msg, msglen, "Invalid %s stride[%d]=0, must be positive", name, i);
      return PFFT_INVALID_CONFIGURATION;
    }
  }
  if (d->number_of_transforms > 1 && distance_of(d, dir) == 0) {
    set_msg(msg, msglen, "Invalid %s distance 0, must be positive for batched FFTs", name);
    return PFFT_INVALID_CONFIGURATION;
  }
  return PFFT_OK;
}

/* descriptor_validation.hpp:123-151 */
static int32_t strides_distance_multidim_check(const pfft_desc_t* d, int dir, const char* name, char* msg,
                                               size_t msglen) {
  uint64_t gs[PFFT_MAX_RANK + 1], gn[PFFT_MAX_RANK + 1];
  int idx[PFFT_MAX_RANK + 1];
  int cnt = 0;
  const uint64_t* strides = strides_of(d, dir);
  for (int i = 0; i < d->rank; ++i) {
    gs[cnt] = strides[i];
    gn[cnt] = d->lengths[i];
    cnt++;
  }
  if (d->number_of_transforms > 1) {
    gs[cnt] = distance_of(d, dir);
    gn[cnt] = d->number_of_transforms;
    cnt++;
  }
  for (int i = 0; i < cnt; ++i) idx[i] = i;
  /* std::sort by stride; insertion sort keeps equal strides in index order */
  for (int i = 1; i < cnt; ++i) {
    int v = idx[i], j = i - 1;
    while (j >= 0 && gs[idx[j]] > gs[v]) {
      idx[j + 1] = idx[j];
      --j;
    }
    idx[j + 1] = v;
  }
  for (int i = 1; i < cnt; ++i) {
    if (!(gs[idx[i - 1]] * gn[idx[i - 1]] <= gs[idx[i]])) {
      set_msg(msg, msglen, "Domain %s: multi-dimension strides are not large enough to avoid overlap", name);
      return PFFT_INVALID_CONFIGURATION;
    }
  }
  return PFFT_OK;
}

/* descriptor_validation.hpp:162-204 */
static int32_t strides_distance_1d_check(const pfft_desc_t* d, int dir, const char* name, char* msg, size_t msglen) {
  const uint64_t fft_size = d->lengths[0];
  const uint64_t stride = strides_of(d, dir)[0];
  const uint64_t distance = distance_of(d, dir);
  const uint64_t nt = d->number_of_transforms;
  const uint64_t first_batch_limit = stride * fft_size;
  const uint64_t first_length_limit = distance * nt;
  if ((stride <= distance && first_batch_limit <= distance) || (distance <= stride && first_length_limit <= stride))
    return PFFT_OK;
  for (uint64_t b = 1; b < nt;) {
    uint64_t batch_first_idx = b * distance;
    uint64_t column = batch_first_idx % stride;
    if (column == 0) {
      if (batch_first_idx >= first_batch_limit) return PFFT_OK;
      set_msg(msg, msglen, "Domain %s: batch %llu collides with first batch at index %llu", name,
              (unsigned long long)b, (unsigned long long)batch_first_idx);
      return PFFT_INVALID_CONFIGURATION;
    }
    uint64_t until = (stride - column) / distance;
    if ((stride - column) % distance != 0) until += 1;
    b += until;
  }
  return PFFT_OK;
}

/* descriptor_validation.hpp:216-225 */
static int32_t strides_distance_check(const pfft_desc_t* d, int dir, const char* name, char* msg, size_t msglen) {
  int32_t st = validate_strides_distance_basic(d, dir, name, msg, msglen);
  if (st != PFFT_OK) return st;
  if (d->rank > 1) return strides_distance_multidim_check(d, dir, name, msg, msglen);
  return strides_distance_1d_check(d, dir, name, msg, msglen);
}

/* descriptor_validation.hpp:264-281 with validate_lengths :38-47, validate_strides_distance :236-253,
 * validate_layout :57-80 */
int32_t pfo_validate(const pfft_desc_t* d, int32_t sg_size, char* msg, size_t msglen) {
  if (d->domain == PFFT_DOMAIN_REAL) {
    set_msg(msg, msglen, "REAL domain is unsupported");
    return PFFT_UNSUPPORTED_CONFIGURATION;
  }
  if (d->number_of_transforms == 0) {
    set_msg(msg, msglen, "Invalid number of transform 0, must be positive");
    return PFFT_INVALID_CONFIGURATION;
  }
  if (d->rank <= 0) {
    set_msg(msg, msglen, "Invalid lengths, must have at least 1 dimension");
    return PFFT_INVALID_CONFIGURATION;
  }
  for (int i = 0; i < d->rank; ++i) {
    if (d->lengths[i] == 0) {
      set_msg(msg, msglen, "Invalid lengths[%d]=0, must be positive", i);
      return PFFT_INVALID_CONFIGURATION;
    }
  }
  int32_t st;
  if (d->placement == PFFT_IN_PLACE) {
    int same = d->n_forward_strides == d->n_backward_strides;
    for (int i = 0; same && i < d->n_forward_strides && i < PFFT_MAX_RANK; ++i)
      same = d->forward_strides[i] == d->backward_strides[i];
    if (!same) {
      set_msg(msg, msglen, "Invalid forward and backward strides must match for in-place configurations");
      return PFFT_INVALID_CONFIGURATION;
    }
    if (d->forward_distance != d->backward_distance) {
      set_msg(msg, msglen, "Invalid forward and backward distances must match for in-place configurations");
      return PFFT_INVALID_CONFIGURATION;
    }
    st = strides_distance_check(d, PFFT_FORWARD, "forward", msg, msglen);
    if (st != PFFT_OK) return st;
  } else {
    st = strides_distance_check(d, PFFT_FORWARD, "forward", msg, msglen);
    if (st != PFFT_OK) return st;
    st = strides_distance_check(d, PFFT_BACKWARD, "backward", msg, msglen);
    if (st != PFFT_OK) return st;
  }
  int fl = pfo_layout(d, PFFT_FORWARD), bl = pfo_layout(d, PFFT_BACKWARD);
  if (d->rank > 1 && !(fl == PFFT_LAYOUT_PACKED && bl == PFFT_LAYOUT_PACKED)) {
    set_msg(msg, msglen, "Multi-dimensional transforms are only supported with default data layout");
    return PFFT_UNSUPPORTED_CONFIGURATION;
  }
  if (fl == PFFT_LAYOUT_UNPACKED || bl == PFFT_LAYOUT_UNPACKED) {
    int scalar_bytes = d->precision == PFFT_PRECISION_F64 ? 8 : 4;
    if (!pfo_fits_in_sg((int64_t)d->lengths[d->rank - 1], scalar_bytes, sg_size)) {
      set_msg(msg, msglen,
              "Arbitrary strides and distances are only supported for sizes that fit in the registers of a subgroup");
      return PFFT_UNSUPPORTED_CONFIGURATION;
    }
  }
  return PFFT_OK;
}

/* ---------------------------------------------------------------------------------------------------------------
 * transforms
 * ------------------------------------------------------------------------------------------------------------- */

#define REAL float
#define FN(x) x##_f32
#include "pfft_oracle_kernels.inc"
#undef REAL
#undef FN
#define REAL double
#define FN(x) x##_f64
#include "pfft_oracle_kernels.inc"
#undef REAL
#undef FN

int32_t pfo_dft_1d(int32_t is_double, int64_t n, int32_t direction, int32_t force_level, int32_t sg_size,
                   int64_t local_mem_bytes, const void* in, void* out, char* msg, size_t msglen) {
  pfo_impl_t impl;
  const int scalar_bytes = is_double ? 8 : 4;
  int32_t st = pfo_prepare_implementation(n, scalar_bytes, sg_size, local_mem_bytes, &impl, msg, msglen);
  if (st != PFFT_OK) return st;
  if (force_level >= 0 && force_level != impl.level) {
    /* re-plan at the requested level when it is feasible */
    memset(&impl, 0, sizeof(impl));
    impl.level = force_level;
    impl.n_kernels = 1;
    impl.kernel_level[0] = force_level;
    impl.kernel_length[0] = n;
    if (force_level == PFO_WORKITEM) {
      if (n > 56) {
        set_msg(msg, msglen, "size %lld cannot be forced onto the work-item level", (long long)n);
        return PFFT_UNSUPPORTED_CONFIGURATION;
      }
    } else if (force_level == PFO_SUBGROUP) {
      int64_t fsg = pfo_factorize_sg(n, sg_size);
      if (n / fsg > 56) {
        set_msg(msg, msglen, "size %lld cannot be forced onto the sub-group level", (long long)n);
        return PFFT_UNSUPPORTED_CONFIGURATION;
      }
      impl.n_factors[0] = 2;
      impl.factors[0][0] = (int32_t)(n / fsg);
      impl.factors[0][1] = (int32_t)fsg;
    } else if (force_level == PFO_WORKGROUP) {
      int64_t nn = pfo_factorize(n), mm = n / nn;
      int64_t sgn = pfo_factorize_sg(nn, sg_size), sgm = pfo_factorize_sg(mm, sg_size);
      if (nn == 1 || nn / sgn > 56 || mm / sgm > 56) {
        set_msg(msg, msglen, "size %lld cannot be forced onto the work-group level", (long long)n);
        return PFFT_UNSUPPORTED_CONFIGURATION;
      }
      impl.n_factors[0] = 4;
      impl.factors[0][0] = (int32_t)(nn / sgn);
      impl.factors[0][1] = (int32_t)sgn;
      impl.factors[0][2] = (int32_t)(mm / sgm);
      impl.factors[0][3] = (int32_t)sgm;
    } else {
      impl.n_kernels = 0;
      int64_t temp = 1;
      /* force a GLOBAL factorisation with a local memory size small enough that nothing larger fits */
      while (n / temp != 1) {
        int64_t f = 0;
        st = factorize_input_impl(n / temp, 1, scalar_bytes, sg_size, local_mem_bytes, &impl, &f, msg, msglen);
        if (st != PFFT_OK) return st;
        temp *= f;
      }
    }
  }
  const int backward = direction == PFFT_BACKWARD;
  if (is_double) {
    double* bx = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* by = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    const double* i = (const double*)in;
    double* o = (double*)out;
    one_transform_f64(i, i + 1, 2, o, o + 1, 2, n, 1, 1, backward, &impl, sg_size, 1.0, 0, bx, by, NULL);
    free(bx);
    free(by);
  } else {
    float* bx = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    float* by = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    const float* i = (const float*)in;
    float* o = (float*)out;
    one_transform_f32(i, i + 1, 2, o, o + 1, 2, n, 1, 1, backward, &impl, sg_size, 1.0f, 0, bx, by, NULL);
    free(bx);
    free(by);
  }
  return PFFT_OK;
}

/* committed_descriptor_impl.hpp:852-950: dispatch_direction + dispatch_dimensions, host version. */
int32_t pfo_compute(const pfft_desc_t* d, int32_t direction, const void* in, void* out, const void* in_imag,
                    void* out_imag, int32_t sg_size, int64_t local_mem_bytes, int32_t n_threads, char* msg,
                    size_t msglen) {
  int32_t st = pfo_validate(d, sg_size, msg, msglen);
  if (st != PFFT_OK) return st;
  const int is_double = d->precision == PFFT_PRECISION_F64;
  const int scalar_bytes = is_double ? 8 : 4;
  const int rank = d->rank;
  pfo_impl_t impls[PFFT_MAX_RANK];
  for (int i = 0; i < rank; ++i) {
    st = pfo_prepare_implementation((int64_t)d->lengths[i], scalar_bytes, sg_size, local_mem_bytes, &impls[i], msg,
                                    msglen);
    if (st != PFFT_OK) return st;
    /* committed_descriptor_impl.hpp:757-764 */
    if (impls[i].level == PFO_GLOBAL) {
      if (rank > 1) {
        set_msg(msg, msglen, "multidimensional global transforms are not supported.");
        return PFFT_UNSUPPORTED_CONFIGURATION;
      }
      if (pfo_layout(d, PFFT_FORWARD) != PFFT_LAYOUT_PACKED || pfo_layout(d, PFFT_BACKWARD) != PFFT_LAYOUT_PACKED) {
        set_msg(msg, msglen, "Large FFTs are currently only supported in non-strided format");
        return PFFT_UNSUPPORTED_CONFIGURATION;
      }
    }
  }
  const int backward = direction == PFFT_BACKWARD;
  const int in_dir = direction, out_dir = backward ? PFFT_FORWARD : PFFT_BACKWARD;
  const uint64_t in_off = offset_of(d, in_dir), out_off = offset_of(d, out_dir);
  const uint64_t in_dist = distance_of(d, in_dir), out_dist = distance_of(d, out_dir);
  const uint64_t* in_str = strides_of(d, in_dir);
  const uint64_t* out_str = strides_of(d, out_dir);
  const double scale_d = backward ? d->backward_scale : d->forward_scale;
  const uint64_t nt = d->number_of_transforms;
  const uint64_t total = pfo_flattened_length(d);
  const int split = d->complex_storage == PFFT_SPLIT_COMPLEX;
  const uint64_t last_len = d->lengths[rank - 1];
  const uint64_t outer_size = total / last_len;
  if (n_threads <= 0) n_threads = 1;
  static_twiddle_init();

  /* element pointers: interleaved -> step 2 through one array; split -> step 1 through two arrays */
  const int step = split ? 1 : 2;
#define RE_PTR(T, base, base_im, idx) (split ? ((T*)(base) + (idx)) : ((T*)(base) + 2 * (idx)))
#define IM_PTR(T, base, base_im, idx) (split ? ((T*)(base_im) + (idx)) : ((T*)(base) + 2 * (idx) + 1))

  /* commit-time twiddles, one set per dimension */
  tables_f32 tabs32[PFFT_MAX_RANK];
  tables_f64 tabs64[PFFT_MAX_RANK];
  for (int i = 0; i < rank; ++i) {
    if (is_double) {
      make_tables_f64(&impls[i], &tabs64[i]);
    } else {
      make_tables_f32(&impls[i], &tabs32[i]);
    }
  }

  /* last dimension: number_of_transforms * outer_size transforms, input layout -> output layout (:923-925).
   * For rank 1 the user strides/distances apply; for rank > 1 the layout is PACKED. */
  {
    const int64_t n = (int64_t)last_len;
    const int64_t count = (int64_t)(nt * outer_size);
    const int apply_scale = 1; /* is_final_factor && is_final_dim (:473-474): the last dimension is run first */
#pragma omp parallel num_threads(n_threads)
    {
      void* bx = malloc((size_t)scalar_bytes * 2 * (size_t)n);
      void* by = malloc((size_t)scalar_bytes * 2 * (size_t)n);
#pragma omp for schedule(static)
      for (int64_t t = 0; t < count; ++t) {
        uint64_t ib, ob, istr, ostr;
        if (rank == 1) {
          ib = in_off + (uint64_t)t * in_dist;
          ob = out_off + (uint64_t)t * out_dist;
          istr = in_str[0];
          ostr = out_str[0];
        } else {
          ib = in_off + (uint64_t)t * last_len;
          ob = out_off + (uint64_t)t * last_len;
          istr = 1;
          ostr = 1;
        }
        if (is_double) {
          one_transform_f64(RE_PTR(const double, in, in_imag, ib), IM_PTR(const double, in, in_imag, ib), step,
                            RE_PTR(double, out, out_imag, ob), IM_PTR(double, out, out_imag, ob), step, n,
                            (int64_t)istr, (int64_t)ostr, backward, &impls[rank - 1], sg_size, (double)scale_d,
                            apply_scale, (double*)bx, (double*)by, &tabs64[rank - 1]);
        } else {
          one_transform_f32(RE_PTR(const float, in, in_imag, ib), IM_PTR(const float, in, in_imag, ib), step,
                            RE_PTR(float, out, out_imag, ob), IM_PTR(float, out, out_imag, ob), step, n,
                            (int64_t)istr, (int64_t)ostr, backward, &impls[rank - 1], sg_size, (float)scale_d,
                            apply_scale, (float*)bx, (float*)by, &tabs32[rank - 1]);
        }
      }
      free(bx);
      free(by);
    }
  }
  /* outer dimensions, in place on `out`, BATCH_INTERLEAVED with stride inner_size (:932-948) */
  uint64_t inner_size = last_len;
  uint64_t outer = outer_size;
  for (int i = rank - 2; i >= 0; --i) {
    outer /= d->lengths[i];
    const int64_t n = (int64_t)d->lengths[i];
    const uint64_t stride_between_kernels = inner_size * d->lengths[i];
    const int64_t count = (int64_t)(nt * outer * inner_size);
    const uint64_t cur_inner = inner_size;
#pragma omp parallel num_threads(n_threads)
    {
      void* bx = malloc((size_t)scalar_bytes * 2 * (size_t)n);
      void* by = malloc((size_t)scalar_bytes * 2 * (size_t)n);
#pragma omp for schedule(static)
      for (int64_t t = 0; t < count; ++t) {
        uint64_t j = (uint64_t)t / cur_inner, c = (uint64_t)t % cur_inner;
        uint64_t base = out_off + j * stride_between_kernels + c;
        if (is_double) {
          one_transform_f64(RE_PTR(const double, out, out_imag, base), IM_PTR(const double, out, out_imag, base), step,
                            RE_PTR(double, out, out_imag, base), IM_PTR(double, out, out_imag, base), step, n,
                            (int64_t)cur_inner, (int64_t)cur_inner, backward, &impls[i], sg_size, 1.0, 0, (double*)bx,
                            (double*)by, &tabs64[i]);
        } else {
          one_transform_f32(RE_PTR(const float, out, out_imag, base), IM_PTR(const float, out, out_imag, base), step,
                            RE_PTR(float, out, out_imag, base), IM_PTR(float, out, out_imag, base), step, n,
                            (int64_t)cur_inner, (int64_t)cur_inner, backward, &impls[i], sg_size, 1.0f, 0, (float*)bx,
                            (float*)by, &tabs32[i]);
        }
      }
      free(bx);
      free(by);
    }
    inner_size *= d->lengths[i];
  }
#undef RE_PTR
#undef IM_PTR
  for (int i = 0; i < rank; ++i) {
    if (is_double) {
      free_tables_f64(&tabs64[i]);
    } else {
      free_tables_f32(&tabs32[i]);
    }
  }
  return PFFT_OK;
}
