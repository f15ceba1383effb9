// XCD-local hand-off yardstick: can the intermediate of a two-stage plan stay in ONE XCD's 4 MiB L2?
//
// profiles/r3_ic_yardstick.txt: two launches through the Infinity Cache top out at 0.41-0.435 of the 8 TB/s peak even
// as plain copies -- every byte crosses the fabric four times (in, intermediate out, intermediate in, out) and the
// fabric moves about 6.7 TB/s whatever the far end is.  An intermediate that never leaves the XCD that produced it
// crosses it twice.  This probe is the copy version of that plan: ONE persistent launch; work-groups read their XCC_ID
// and take tasks from a per-XCD ticket counter; unit u (one "FFT" of U bytes) belongs to XCD u % 8; phase p of an XCD
// holds the stage-A tasks of its unit p (IN -> scratch slot p % SLOTS, plain stores: dirty lines stay in this L2) and the
// stage-B tasks of unit p - 1 (scratch -> OUT; every B task reads a piece of EVERY A tile, as a transposing four-step
// stage B does; sc1 loads: L1 bypassed, served by the XCD's L2).  Dependencies through agent-scope counters
// (MI355X_MICROARCH.md, "inter-workgroup visibility").  Result checked bit for bit against the same permutation done on
// the host side index map.
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/xcd_local.hip -o build/xcd_local
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr int AUX_DEF = 0, AUX_NT = 2, AUX_SC1 = 0x10;
constexpr int MAX_XCD = 8;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__global__ void fill_random(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + seed * 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
    unsigned m = (unsigned)(z >> 40) | 0x3F800000u;
    p[i] = __float_as_uint((__uint_as_float(m) - 1.5f) * 2.0f);
  }
}

// Unit geometry: R rows x (G * SEG) bytes; A group g = column block g (SEG bytes x R rows) -> tile g (R * SEG contiguous);
// B group j = rows [j * R / G, (j + 1) * R / G) of every tile -> output column block j of the "transposed" unit:
// OUT row (g * R / G + r) , bytes [j * SEG, (j + 1) * SEG)   (a permutation of 16-byte words; checked by the host)
template <int SEG, int R, int G, int WG>
struct geom {
  static constexpr int LPS = SEG / 16;       // lanes per segment
  static constexpr int RPI = WG / LPS;       // rows per wave-instruction
  static constexpr int IT = R / RPI;         // 16-byte loads per lane
  static constexpr unsigned PITCH = G * SEG;
  static constexpr unsigned TILE = R * SEG;
  static constexpr unsigned UNIT = (unsigned)R * PITCH;
  static constexpr int RB = R / G;           // rows of a tile that one B group takes
  static_assert(RB % RPI == 0 || RPI % RB == 0, "shape");
};

struct ctl_t {                 // per XCD, 128-byte separated counters
  unsigned ticket; unsigned pad0[31];
  unsigned timeouts; unsigned pad1[31];
};
// bounded spin (a probe must not hang the box): gives up after ~50 ms and counts it
__device__ __forceinline__ void spin_until(unsigned* p, unsigned want, unsigned* timeouts) {
  for (unsigned n = 0; __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++n) {
    __builtin_amdgcn_s_sleep(2);
    if (n > (1u << 14) || __hip_atomic_load(timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { atomicAdd(timeouts, 1u); break; }
  }
}

template <int SEG, int R, int G, int WG, int SLOTS, int ST_MID>
__device__ __noinline__ void task_a(const char* in, char* my_scratch, unsigned* da, unsigned* db, unsigned* timeouts, unsigned x, int k, int g, int n_xcd) {
  using Gm = geom<SEG, R, G, WG>;
  const int tid = threadIdx.x;
  const unsigned r0 = tid / Gm::LPS, b0 = (tid % Gm::LPS) * 16;
  const size_t unit = (size_t)x + (size_t)k * n_xcd;
  char* const slot = my_scratch + (size_t)(k % SLOTS) * Gm::UNIT;
  v4u v[Gm::IT];
  auto rs = rsrc_of(in + unit * Gm::UNIT, Gm::UNIT);
#pragma unroll
  for (int i = 0; i < Gm::IT; ++i)
    v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, r0 * Gm::PITCH + g * SEG + b0, (unsigned)(i * Gm::RPI) * Gm::PITCH, AUX_NT);
  // the slot must have been drained by the B tasks of unit k - SLOTS
  if (k >= SLOTS) {
    if (tid == 0) spin_until(&db[k - SLOTS], (unsigned)G, timeouts);
    __syncthreads();
  }
  auto rd = rsrc_of(slot + (size_t)g * Gm::TILE, Gm::TILE);
#pragma unroll
  for (int i = 0; i < Gm::IT; ++i)
    __builtin_amdgcn_raw_buffer_store_b128(v[i], rd, r0 * SEG + b0, (unsigned)(i * Gm::RPI * SEG), ST_MID);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(&da[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int SEG, int R, int G, int WG, int SLOTS, int LD_MID>
__device__ __noinline__ void task_b(char* out, char* my_scratch, unsigned* da, unsigned* db, unsigned* timeouts, unsigned x, int k, int j, int n_xcd) {
  using Gm = geom<SEG, R, G, WG>;
  const int tid = threadIdx.x;
  const unsigned r0 = tid / Gm::LPS, b0 = (tid % Gm::LPS) * 16;
  const size_t unit = (size_t)x + (size_t)k * n_xcd;
  char* const slot = my_scratch + (size_t)(k % SLOTS) * Gm::UNIT;
  v4u v[Gm::IT];
  if (tid == 0) spin_until(&da[k], (unsigned)G, timeouts);
  __syncthreads();
  auto rs = rsrc_of(slot, Gm::UNIT);
  // rows [j * RB, (j + 1) * RB) of every tile: iteration i covers tile (i * RPI) / RB, rows (i * RPI) % RB + r0 ...
#pragma unroll
  for (int i = 0; i < Gm::IT; ++i) {
    const unsigned row = (unsigned)(i * Gm::RPI) + r0;   // 0 .. R - 1 over (tile, row-in-block)
    const unsigned tile = row / Gm::RB, rr = row % Gm::RB;
    v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, tile * Gm::TILE + ((unsigned)j * Gm::RB + rr) * SEG + b0, 0, LD_MID);
  }
  auto rd = rsrc_of(out + unit * Gm::UNIT, Gm::UNIT);
#pragma unroll
  for (int i = 0; i < Gm::IT; ++i)
    __builtin_amdgcn_raw_buffer_store_b128(v[i], rd, r0 * Gm::PITCH + j * SEG + b0, (unsigned)(i * Gm::RPI) * Gm::PITCH, AUX_NT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loads of the slot have returned
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(&db[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int SEG, int R, int G, int WG, int SLOTS, int LD_MID, int ST_MID>
__global__ __launch_bounds__(WG) void xcd_pipeline(const char* in, char* out, char* scratch, ctl_t* ctl, unsigned* done_a,
                                                  unsigned* done_b, int units, int n_xcd, int max_k) {
  using Gm = geom<SEG, R, G, WG>;
  __shared__ unsigned s_ticket;
  const unsigned x = xcc_id() % n_xcd;
  const int tid = threadIdx.x;
  const int K = (units - (int)x + n_xcd - 1) / n_xcd;  // units of this XCD: x, x + n_xcd, ...
  char* const my_scratch = scratch + (size_t)x * SLOTS * Gm::UNIT;
  unsigned* const da = done_a + (size_t)x * max_k;
  unsigned* const db = done_b + (size_t)x * max_k;
  // (no continue / break in the middle of the loop body: the barriers sit in straight-line uniform code)
  bool more = true;
  for (unsigned iter = 0; more && iter < (1u << 20); ++iter) {
    if (tid == 0) s_ticket = atomicAdd(&ctl[x].ticket, 1u);
    __syncthreads();
    const unsigned t = __builtin_amdgcn_readfirstlane(s_ticket);
    __syncthreads();
    const int p = (int)(t / (2 * G)), q = (int)(t % (2 * G));
    const bool is_a = q < G;
    const int k = is_a ? p : p - 1;  // local unit index
    more = p <= K;
    if (more && k >= 0 && k < K) {
      if (is_a) task_a<SEG, R, G, WG, SLOTS, ST_MID>(in, my_scratch, da, db, &ctl[x].timeouts, x, k, q, n_xcd);
      else task_b<SEG, R, G, WG, SLOTS, LD_MID>(out, my_scratch, da, db, &ctl[x].timeouts, x, k, q - G, n_xcd);
    }
  }
}

// the same permutation as two ordinary launches through a scratch of `chunk_units` units (the production structure)
template <int SEG, int R, int G, int WG, int ST_MID, int LD_MID>
__global__ __launch_bounds__(WG) void stage_kernel(const char* in, char* out, int units, int stage) {
  using Gm = geom<SEG, R, G, WG>;
  const int tid = threadIdx.x;
  const unsigned r0 = tid / Gm::LPS, b0 = (tid % Gm::LPS) * 16;
  for (long long task = blockIdx.x; task < (long long)units * G; task += gridDim.x) {
    const size_t unit = task / G; const int g = (int)(task % G);
    v4u v[Gm::IT];
    auto rs = rsrc_of(in + unit * Gm::UNIT, Gm::UNIT);
    auto rd = rsrc_of(out + unit * Gm::UNIT, Gm::UNIT);
    if (stage == 0) {
#pragma unroll
      for (int i = 0; i < Gm::IT; ++i)
        v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, r0 * Gm::PITCH + g * SEG + b0, (unsigned)(i * Gm::RPI) * Gm::PITCH, AUX_NT);
#pragma unroll
      for (int i = 0; i < Gm::IT; ++i)
        __builtin_amdgcn_raw_buffer_store_b128(v[i], rd, (unsigned)g * Gm::TILE + r0 * SEG + b0, (unsigned)(i * Gm::RPI * SEG), ST_MID);
    } else {
#pragma unroll
      for (int i = 0; i < Gm::IT; ++i) {
        const unsigned row = (unsigned)(i * Gm::RPI) + r0;
        const unsigned tile = row / Gm::RB, rr = row % Gm::RB;
        v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, tile * Gm::TILE + ((unsigned)g * Gm::RB + rr) * SEG + b0, 0, LD_MID);
      }
#pragma unroll
      for (int i = 0; i < Gm::IT; ++i)
        __builtin_amdgcn_raw_buffer_store_b128(v[i], rd, r0 * Gm::PITCH + g * SEG + b0, (unsigned)(i * Gm::RPI) * Gm::PITCH, AUX_NT);
    }
  }
}

__global__ void xcc_map(unsigned* o) { if (threadIdx.x == 0) { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); o[blockIdx.x] = v; } }

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <int SEG, int R, int G, int WG, int SLOTS>
void run(const char* name, char* in, char* out, char* ref, char* scratch, size_t total, int wg_per_cu, int cus) {
  using Gm = geom<SEG, R, G, WG>;
  const int units = (int)(total / Gm::UNIT);
  const int n_xcd = 8, max_k = units / n_xcd + 2;
  ctl_t* ctl; unsigned *da, *db;
  CK(hipMalloc(&ctl, sizeof(ctl_t) * MAX_XCD)); CK(hipMalloc(&da, sizeof(unsigned) * MAX_XCD * max_k)); CK(hipMalloc(&db, sizeof(unsigned) * MAX_XCD * max_k));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("---- %s: unit %u KiB = %d rows x %d B, %d groups of %d B segments, %d lanes, %d units, %d slots per XCD\n", name,
         Gm::UNIT >> 10, R, (int)Gm::PITCH, G, SEG, WG, units, SLOTS);
  // reference: two launches through a full-size scratch (= `ref` holds the result)
  {
    std::vector<double> t;
    const size_t chunk = (size_t)256 << 20; const int cu_units = (int)(chunk / Gm::UNIT);
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      for (int u0 = 0; u0 < units; u0 += cu_units) {
        const int nu = std::min(cu_units, units - u0);
        hipLaunchKernelGGL((stage_kernel<SEG, R, G, WG, AUX_SC1, AUX_DEF>), dim3(nu * G / 4), dim3(WG), 0, 0, in + (size_t)u0 * Gm::UNIT, scratch, nu, 0);
        hipLaunchKernelGGL((stage_kernel<SEG, R, G, WG, AUX_SC1, AUX_DEF>), dim3(nu * G / 4), dim3(WG), 0, 0, scratch, ref + (size_t)u0 * Gm::UNIT, nu, 1);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) t.push_back(ms);
    }
    printf("two launches per 256 MiB chunk (Infinity Cache):        %.3f ms = %.3f of 8 TB/s on 1x bytes\n", median(t), 2.0 * total * 1e-9 / median(t) / 8.0);
  }
  auto one = [&](const char* what, auto kernel, int grid) {
    std::vector<double> t;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemsetAsync(ctl, 0, sizeof(ctl_t) * MAX_XCD)); CK(hipMemsetAsync(da, 0, sizeof(unsigned) * MAX_XCD * max_k)); CK(hipMemsetAsync(db, 0, sizeof(unsigned) * MAX_XCD * max_k));
      if (rep == 0) CK(hipMemsetAsync(out, 0, total));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(WG), 0, 0, in, out, scratch, ctl, da, db, units, n_xcd, max_k);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) t.push_back(ms);
      if (rep == 0) {  // verify against the two-launch result
        std::vector<unsigned> a(1 << 20), b(1 << 20);
        size_t bad = 0;
        for (size_t off : {(size_t)0, total / 2, total - ((size_t)4 << 20)}) {  // total >= 8 MiB
          CK(hipMemcpy(a.data(), out + off, 4 << 20, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), ref + off, 4 << 20, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
        }
        ctl_t h[MAX_XCD]; CK(hipMemcpy(h, ctl, sizeof h, hipMemcpyDeviceToHost));
        unsigned to = 0; for (auto& c : h) to += c.timeouts;
        if (bad || to) printf("   !! %zu mismatching words, %u spin timeouts\n", bad, to);
      }
    }
    printf("%-56s%.3f ms = %.3f of 8 TB/s on 1x bytes (grid %d)\n", what, median(t), 2.0 * total * 1e-9 / median(t) / 8.0, grid);
  };
  for (int w : {wg_per_cu, wg_per_cu * 2}) {
    char buf[128];
    snprintf(buf, sizeof buf, "XCD-local, plain stores / sc1 loads, %d WG per CU:", w);
    one(buf, xcd_pipeline<SEG, R, G, WG, SLOTS, AUX_SC1, AUX_DEF>, cus * w);
    snprintf(buf, sizeof buf, "XCD-local, plain stores / nt loads, %d WG per CU:", w);
    one(buf, xcd_pipeline<SEG, R, G, WG, SLOTS, AUX_NT, AUX_DEF>, cus * w);
    snprintf(buf, sizeof buf, "XCD-local, sc1 stores / sc1 loads (L2 dropped), %d WG/CU:", w);
    one(buf, xcd_pipeline<SEG, R, G, WG, SLOTS, AUX_SC1, AUX_SC1>, cus * w);
  }
  CK(hipFree(ctl)); CK(hipFree(da)); CK(hipFree(db));
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t total = (size_t)2 << 30;
  char *in, *out, *ref, *scratch;
  CK(hipMalloc(&in, total)); CK(hipMalloc(&out, total)); CK(hipMalloc(&ref, total)); CK(hipMalloc(&scratch, (size_t)256 << 20));
  fill_random<<<4096, 256>>>((unsigned*)in, total / 4, 1);
  CK(hipDeviceSynchronize());
  {
    unsigned* d; CK(hipMalloc(&d, 4096 * 4)); xcc_map<<<4096, 64>>>(d); std::vector<unsigned> h(4096); CK(hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost));
    printf("raw HW_REG_XCC_ID of blocks 0..23:"); for (int i = 0; i < 24; ++i) printf(" %x", h[i]); printf("\n");
    int cnt[16] = {0}, rr = 0; for (int i = 0; i < 4096; ++i) { cnt[h[i] & 15]++; rr += (h[i] & 15) == (h[i % 8] & 15); }
    printf("blocks per XCC_ID & 15:"); for (int i = 0; i < 16; ++i) printf(" %d", cnt[i]); printf("; blocks b with xcc(b) == xcc(b %% 8): %d of 4096\n", rr);
  }
  run<256, 256, 8, 512, 3>("small sanity run", in, out, ref, scratch, (size_t)64 << 20, 2, cus);
  // fp32 N = 65536 = 256 x 256: unit 512 KiB, 8 groups of 32 columns (256-byte segments) x 256 rows
  run<256, 256, 8, 512, 3>("fp32 65536 shape", in, out, ref, scratch, total, 2, cus);
  run<256, 256, 8, 512, 2>("fp32 65536 shape", in, out, ref, scratch, total, 2, cus);
  // fp32 N = 2^17 = 512 x 256 -> unit 1 MiB: 512 rows x 2048 B
  run<256, 512, 8, 512, 2>("1 MiB unit (fp32 2^17 / fp64 2^16)", in, out, ref, scratch, total, 1, cus);
  // fp32 N = 2^18 = 512 x 512: unit 2 MiB, 16 groups of 256 B x 512 rows
  run<256, 512, 16, 512, 2>("2 MiB unit (fp32 2^18)", in, out, ref, scratch, total, 1, cus);
  // fp64 N = 32768 = 256 x 128 -> unit 512 KiB with 128-byte segments: 256 rows x 2048 B, 16 groups
  run<128, 256, 16, 256, 3>("512 KiB unit, 128 B segments (fp64 32768)", in, out, ref, scratch, total, 2, cus);
  CK(hipGetLastError());
  return 0;
}
