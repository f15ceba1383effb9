/*
 * portfft_amd.h -- C ABI of the MI355X-native batched FFT engine.
 *
 * This is the drop-in boundary for the 1-D / N-D complex-to-complex execute path of portFFT.  Every entry point
 * names the reference interface it replaces (paths relative to /root/reference).  The C++ facade
 * include/portfft/portfft.hpp (namespace portfft: descriptor, committed_descriptor, exceptions) and the Python
 * mirror portfft_amd/ are thin wrappers over exactly these symbols.
 *
 * Conventions
 *   - plain C types only: pointers, sizes, enums as int32_t; no HIP or torch types in signatures
 *     (a HIP stream is passed as void*; NULL = the default stream).
 *   - every function returns a pfft_status; the message of the last failure on the calling thread is
 *     available from pfft_last_error().  No C++ exception crosses this boundary.
 *   - `in`/`out` are device-accessible pointers (hipMalloc / hipMallocManaged), i.e. the USM pointers of the
 *     reference's compute_forward/compute_backward overloads.
 *   - execution is asynchronous and ordered on the plan's stream, like the reference's sycl::event-returning
 *     overloads (src/portfft/committed_descriptor.hpp:171-310).
 */
#ifndef PORTFFT_AMD_H
#define PORTFFT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFFT_MAX_RANK 8

/* Error taxonomy: mirrors src/portfft/common/exceptions.hpp:32-77. */
typedef enum pfft_status {
  PFFT_OK = 0,
  PFFT_INVALID_CONFIGURATION = 1,     /* portfft::invalid_configuration      */
  PFFT_UNSUPPORTED_CONFIGURATION = 2, /* portfft::unsupported_configuration  */
  PFFT_OUT_OF_LOCAL_MEMORY = 3,       /* portfft::out_of_local_memory_error  */
  PFFT_INTERNAL_ERROR = 4,            /* portfft::internal_error             */
  PFFT_HIP_ERROR = 5                  /* a HIP runtime call failed (reference: sycl::exception) */
} pfft_status;

/* src/portfft/enums.hpp:25-31 */
enum { PFFT_DOMAIN_REAL = 0, PFFT_DOMAIN_COMPLEX = 1 };
enum { PFFT_INTERLEAVED_COMPLEX = 0, PFFT_SPLIT_COMPLEX = 1 };
enum { PFFT_IN_PLACE = 0, PFFT_OUT_OF_PLACE = 1 };
enum { PFFT_FORWARD = 0, PFFT_BACKWARD = 1 };
/* src/portfft/enums.hpp:44-56 (detail::layout) */
enum { PFFT_LAYOUT_PACKED = 0, PFFT_LAYOUT_UNPACKED = 1, PFFT_LAYOUT_BATCH_INTERLEAVED = 2 };
enum { PFFT_PRECISION_F32 = 0, PFFT_PRECISION_F64 = 1 };

/*
 * POD mirror of portfft::descriptor<Scalar, Domain> (src/portfft/descriptor.hpp:43-129): same fields, same
 * meaning, same defaults (set by pfft_desc_init).  `n_forward_strides`/`n_backward_strides` carry the vector
 * sizes so that the "mismatching strides length" check of descriptor_validation.hpp:92-99 can be reproduced.
 */
typedef struct pfft_desc_t {
  int32_t precision;       /* PFFT_PRECISION_*: the Scalar template argument */
  int32_t domain;          /* PFFT_DOMAIN_*: the Domain template argument */
  int32_t rank;            /* lengths.size() */
  int32_t complex_storage; /* PFFT_INTERLEAVED_COMPLEX (default) | PFFT_SPLIT_COMPLEX */
  int32_t placement;       /* PFFT_OUT_OF_PLACE (default) | PFFT_IN_PLACE */
  int32_t n_forward_strides;
  int32_t n_backward_strides;
  int32_t reserved_;
  uint64_t lengths[PFFT_MAX_RANK];
  uint64_t forward_strides[PFFT_MAX_RANK];
  uint64_t backward_strides[PFFT_MAX_RANK];
  uint64_t forward_distance;
  uint64_t backward_distance;
  uint64_t forward_offset;
  uint64_t backward_offset;
  uint64_t number_of_transforms;
  double forward_scale;
  double backward_scale;
} pfft_desc_t;

/* Tier a dimension was planned on; the analogue of detail::level (src/portfft/enums.hpp:42). */
enum {
  PFFT_TIER_REGISTER = 0,  /* one register pass per FFT (reference: WORKITEM) */
  PFFT_TIER_WORKGROUP = 1, /* Stockham passes through LDS, specialised kernel (reference: SUBGROUP + WORKGROUP) */
  PFFT_TIER_GENERIC = 2,   /* runtime-radix LDS kernel, any stride / storage */
  PFFT_TIER_GLOBAL = 3     /* multi-kernel decomposition through HBM scratch (reference: GLOBAL) */
};

#define PFFT_MAX_FACTORS 16
typedef struct pfft_dim_info_t {
  uint64_t length;
  int32_t tier;
  int32_t n_factors;
  int32_t factors[PFFT_MAX_FACTORS]; /* radices of the passes (WORKGROUP/GENERIC) or sub-lengths (GLOBAL) */
  int32_t workgroup_size;
  int32_t ffts_per_workgroup;
  uint64_t lds_bytes;
} pfft_dim_info_t;

typedef struct pfft_plan_info_t {
  int32_t rank;
  int32_t n_compute_units;
  uint64_t twiddle_bytes; /* HBM held by the plan for twiddles */
  uint64_t scratch_bytes; /* HBM held by the plan for intermediate data */
  pfft_dim_info_t dims[PFFT_MAX_RANK];
  int32_t launches[2]; /* kernel launches of one execute [forward, backward]; plans that run chunk by chunk
                          (intermediate sized to the Infinity Cache) count every chunk's launches */
  int32_t xcd_local[2]; /* 1: that direction runs the GLOBAL tier as ONE persistent launch (per-XCD task queues) followed
                           by its recovery launch, which does nothing unless a hand-off wait of the former gave up */
  uint64_t xcd_recoveries; /* executes of this plan (copy) whose persistent launch gave up and were recomputed, in stream
                              order, by the recovery launch: the result was valid whenever the execute's event completed */
  uint64_t knob_mask; /* which PFFT_* environment knobs differed from their defaults when the plan was committed (bit order:
                         plan_knobs::from_env, portfft_amd/csrc/plan_core.cpp); 0 = the product's defaults.  The environment
                         is read at commit only, never at execute */
} pfft_plan_info_t;

typedef struct pfft_plan_t pfft_plan_t; /* opaque: portfft::committed_descriptor */

/* ---- descriptor (host only, no device needed) ------------------------------------------------------------ */

/* descriptor::descriptor(lengths): default strides / distances / scales (src/portfft/descriptor.hpp:131-144). */
pfft_status pfft_desc_init(pfft_desc_t* desc, int32_t precision, int32_t domain, int32_t rank,
                           const uint64_t* lengths);
/* detail::validate::validate_descriptor (src/portfft/descriptor_validation.hpp:264-281). */
pfft_status pfft_desc_validate(const pfft_desc_t* desc);
/* descriptor::get_flattened_length (src/portfft/descriptor.hpp:161-163). */
uint64_t pfft_desc_flattened_length(const pfft_desc_t* desc);
/* descriptor::get_input_count / get_output_count (src/portfft/descriptor.hpp:172-183). */
uint64_t pfft_desc_input_count(const pfft_desc_t* desc, int32_t direction);
uint64_t pfft_desc_output_count(const pfft_desc_t* desc, int32_t direction);
/* detail::get_layout (src/portfft/utils.hpp:238-246). */
int32_t pfft_desc_layout(const pfft_desc_t* desc, int32_t direction);

/* ---- plan (needs a HIP device) ----------------------------------------------------------------------------- */

/* descriptor::commit(queue) (src/portfft/descriptor.hpp:152-156): validate, plan every dimension, upload
 * twiddles, allocate scratch.  `hip_stream` is a hipStream_t (NULL = default stream) on the current device.
 * Configurations without a pre-compiled kernel are specialised here by hiprtc (0.2-2 s the first time, then cached
 * in the process and on disk; the reference builds its kernels at commit too: committed_descriptor_impl.hpp:448-573);
 * nothing is compiled or allocated at execute.  Thread-safe; the plan itself is not (one plan per host thread). */
pfft_status pfft_plan_create(const pfft_desc_t* desc, void* hip_stream, pfft_plan_t** plan);
/* committed_descriptor_impl::~committed_descriptor_impl (committed_descriptor_impl.hpp:825-828): waits for the
 * stream, frees twiddles and scratch. */
pfft_status pfft_plan_destroy(pfft_plan_t* plan);
/* Planner output for tests / logging (no reference equivalent beyond PORTFFT_LOG_TRACE). */
pfft_status pfft_plan_get_info(const pfft_plan_t* plan, pfft_plan_info_t* info);

/* committed_descriptor::compute_forward / compute_backward, interleaved USM overloads
 * (src/portfft/committed_descriptor.hpp:171-176, 215-218, 242-246, 288-293).  in == out selects the in-place
 * overload.  `direction` is PFFT_FORWARD or PFFT_BACKWARD.  Storage mismatch -> PFFT_INVALID_CONFIGURATION like
 * dispatch_direction (committed_descriptor_impl.hpp:862-871). */
pfft_status pfft_execute(pfft_plan_t* plan, int32_t direction, const void* in, void* out);
/* Split-complex USM overloads (src/portfft/committed_descriptor.hpp:186-192, 228-232, 258-263, 305-310). */
pfft_status pfft_execute_split(pfft_plan_t* plan, int32_t direction, const void* in_real, const void* in_imag,
                               void* out_real, void* out_imag);
/* The same overloads with the reference's `const std::vector<sycl::event>& dependencies` argument and its returned
 * sycl::event (src/portfft/committed_descriptor.hpp:171, 215, 242-246, 288-293; split twins 186-192, 228-232,
 * 258-263, 305-310).  `deps` is an array of `n_deps` hipEvent_t handles (as void*; NULL entries are skipped): the
 * plan's stream waits for each of them (hipStreamWaitEvent) before the first kernel.  When `event_out` is not NULL
 * it receives a fresh hipEvent_t recorded behind the last kernel of THIS submission; the caller owns it
 * (pfft_event_wait / pfft_event_query / pfft_event_destroy, or any HIP call that takes a hipEvent_t). */
pfft_status pfft_execute_ex(pfft_plan_t* plan, int32_t direction, const void* in, void* out, int32_t n_deps,
                            void* const* deps, void** event_out);
pfft_status pfft_execute_split_ex(pfft_plan_t* plan, int32_t direction, const void* in_real, const void* in_imag,
                                  void* out_real, void* out_imag, int32_t n_deps, void* const* deps,
                                  void** event_out);
/* sycl::event::wait() / get_info<command_execution_status>() / destruction of an event returned by the _ex calls. */
pfft_status pfft_event_wait(void* event);
pfft_status pfft_event_query(void* event, int32_t* done);
pfft_status pfft_event_destroy(void* event);
/* committed_descriptor's copy constructor / copy assignment (committed_descriptor_impl.hpp:774-817): the copy shares
 * the kernels and twiddle tables and gets scratch buffers of its own. */
pfft_status pfft_plan_clone(const pfft_plan_t* plan, pfft_plan_t** copy);
/* sycl::queue::copy(src, dest, count, dependencies) and sycl::queue::wait() as the reference's callers use them around
 * compute_* (test/unit_test/fft_test_utils.hpp:286-333, test/bench/portfft/launch_bench.hpp:96-135): an asynchronous
 * copy of `bytes` bytes on `hip_stream` (host or device pointers) behind `deps`, with its own completion event. */
pfft_status pfft_queue_copy(void* hip_stream, const void* src, void* dst, size_t bytes, int32_t n_deps,
                            void* const* deps, void** event_out);
pfft_status pfft_queue_wait(void* hip_stream);
/* Blocks until everything queued on the plan's stream is done (queue.wait() of the reference's destructor path). */
pfft_status pfft_plan_wait(pfft_plan_t* plan);

/* ---- misc ----------------------------------------------------------------------------------------------------- */
const char* pfft_last_error(void);
const char* pfft_status_string(pfft_status s);
/* "portfft_amd x.y (gfx950)" */
const char* pfft_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PORTFFT_AMD_H */
