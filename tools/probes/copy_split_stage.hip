// Yardstick for the four-step stages of SPLIT_COMPLEX data: the user-facing side is two planes addressed column-shaped
// in segments of COLS scalars (fp32: 16 columns = 64 bytes, 32 columns = 128 bytes) x ROWS rows at a pitch of `pitch`
// scalars, the other side the interleaved group-major scratch (one contiguous tile per work-group), through a chunk of
// 256 MiB like the plan.  Stage A: planes -> tiles (nt loads, default stores); stage B: tiles -> planes (default loads,
// nt stores).  Reported: fraction of 8 TB/s on 1x bytes of the pair, as bench.py counts.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/copy_split_stage.hip -o build/copy_split_stage
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

// group g of a transform = columns [g * COLS, (g + 1) * COLS) x ROWS rows; lanes: column fastest
template <int COLS, int ROWS, int WG, bool TO_TILES>
__global__ __launch_bounds__(WG) void stage(const float* re, const float* im, v2f* tiles, float* ore, float* oim, long long groups,
                                           long long pitch, long long per_transform_groups) {
  constexpr int RPI = WG / COLS, IT = ROWS / RPI;
  const int c = threadIdx.x % COLS, r0 = threadIdx.x / COLS;
  for (long long g = blockIdx.x; g < groups; g += gridDim.x) {
    const long long t = g / per_transform_groups, cg = g % per_transform_groups;
    const long long pbase = t * ROWS * pitch + cg * COLS + c;
    v2f v[IT];
    if (TO_TILES) {
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const long long o = pbase + (long long)(r0 + i * RPI) * pitch;
        v[i].x = __builtin_nontemporal_load(re + o);
        v[i].y = __builtin_nontemporal_load(im + o);
      }
#pragma unroll
      for (int i = 0; i < IT; ++i) tiles[g * (long long)(ROWS * COLS) + (r0 + i * RPI) * COLS + c] = v[i];
    } else {
#pragma unroll
      for (int i = 0; i < IT; ++i) v[i] = tiles[g * (long long)(ROWS * COLS) + (r0 + i * RPI) * COLS + c];
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const long long o = pbase + (long long)(r0 + i * RPI) * pitch;
        __builtin_nontemporal_store(v[i].x, ore + o);
        __builtin_nontemporal_store(v[i].y, oim + o);
      }
    }
  }
}

template <int COLS, int ROWS, int WG>
void run(const char* name, float* re, float* im, float* ore, float* oim, v2f* tiles, size_t elems, int cus) {
  const long long pitch = ROWS;  // square split: n1 = n2 = ROWS, the planes of one transform are ROWS x ROWS
  const long long per_t = pitch / COLS, transforms = (long long)(elems / ((size_t)ROWS * pitch));
  const long long chunk_t = std::max<long long>(1, (((size_t)256 << 20) / 8) / ((size_t)ROWS * pitch));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int r = 0; r < 6; ++r) {
    CK(hipEventRecord(e0));
    for (long long t0 = 0; t0 < transforms; t0 += chunk_t) {
      const long long nt = std::min(chunk_t, transforms - t0), groups = nt * per_t;
      const long long off = t0 * ROWS * pitch;
      const unsigned grid = (unsigned)std::min<long long>(groups, 8LL * cus);
      hipLaunchKernelGGL((stage<COLS, ROWS, WG, true>), dim3(grid), dim3(WG), 0, 0, re + off, im + off, tiles, nullptr, nullptr, groups, pitch, per_t);
      hipLaunchKernelGGL((stage<COLS, ROWS, WG, false>), dim3(grid), dim3(WG), 0, 0, nullptr, nullptr, tiles, ore + off, oim + off, groups, pitch, per_t);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  const double bytes = 2.0 * elems * 8;
  printf("%-58s %.3f ms = %.3f of 8 TB/s on 1x bytes\n", name, t[t.size() / 2], bytes / (t[t.size() / 2] * 1e-3) / 8e12);
  CK(hipGetLastError());
}

int main() {
  const size_t elems = (size_t)1 << 27;  // complex elements: 2 x 512 MiB of planes in, the same out
  float *re, *im, *ore, *oim; v2f* tiles;
  CK(hipMalloc(&re, elems * 4)); CK(hipMalloc(&im, elems * 4)); CK(hipMalloc(&ore, elems * 4)); CK(hipMalloc(&oim, elems * 4));
  CK(hipMalloc(&tiles, (size_t)256 << 20));
  CK(hipMemset(re, 1, elems * 4)); CK(hipMemset(im, 2, elems * 4));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); const int cus = prop.multiProcessorCount;
  run<16, 1024, 1024>("fp32 2^20 split: 16-column plane segments (64 B) x 1024 rows", re, im, ore, oim, tiles, elems, cus);
  run<32, 1024, 1024>("fp32 2^20 split: 32-column plane segments (128 B) x 1024 rows", re, im, ore, oim, tiles, elems, cus);
  run<16, 256, 256>("fp32 65536 split: 16-column plane segments x 256 rows", re, im, ore, oim, tiles, elems, cus);
  run<32, 256, 512>("fp32 65536 split: 32-column plane segments x 256 rows", re, im, ore, oim, tiles, elems, cus);
  run<64, 256, 1024>("fp32 65536 split: 64-column plane segments x 256 rows", re, im, ore, oim, tiles, elems, cus);
  return 0;
}
