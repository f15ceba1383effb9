// 64-byte segments (fp32 n = 2048 stages: 8 columns; SPLIT_COMPLEX planes at 16 columns): does it help to run the two
// work-groups that own the two halves of the same 128-byte lines on the SAME XCD at about the same time?
// Copy kernel, strided in (SEG bytes x ROWS rows at a row pitch) -> contiguous out and the reverse; group -> work-group
// mappings: 0 = grid-stride (group g on block g % grid: neighbours land on different XCDs), 1 = paired (blocks b and
// b + 8 -- same XCD, dispatched back to back -- take groups 2k and 2k + 1), 2 = one block takes both halves in turn.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/seg64_pairing.hip -o build/seg64_pairing
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__global__ void fill_random(unsigned* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull; z ^= z >> 31; z *= 0x94D049BB133111EBull;
    p[i] = 0x3f000000u | (unsigned)(z >> 41);
  }
}
// BPL bytes per lane (8: fp32 complex element, 4: one fp32 plane value is too narrow for this probe -> 8 and 16 only)
template <int SEG, int ROWS, int WG, int BPL, int MAP, bool OUT_STRIDED>
__global__ __launch_bounds__(WG) void copy_cols(const char* in, char* out, long long groups, long long pitch) {
  constexpr int LPS = SEG / BPL, RPI = WG / LPS, IT = ROWS / RPI;
  using vec = typename std::conditional<BPL == 16, v4u, v2u>::type;
  const int tid = threadIdx.x;
  const long long gpi = pitch / SEG;
  const long long nblk = gridDim.x;
  for (long long it = 0;; ++it) {
    long long g;
    if (MAP == 0) g = blockIdx.x + it * nblk;
    else if (MAP == 1) {  // blocks b and b + 8 share an XCD: they take neighbouring groups
      const long long b = blockIdx.x, pairbase = (b / 16) * 16, x = b % 8, h = (b / 8) % 2;
      g = pairbase + 2 * x + h + it * nblk;
    } else {  // MAP 2: one block takes groups 2k, 2k + 1 in turn
      g = 2 * (blockIdx.x + (it / 2) * nblk) + (it & 1);
    }
    if (g >= groups) break;
    const long long img = g / gpi, cg = g % gpi;
    const long long img_bytes = (long long)ROWS * pitch;
    auto rs = rsrc_of(in + img * img_bytes, (unsigned)img_bytes);
    auto rd = rsrc_of(out + img * img_bytes, (unsigned)img_bytes);
    const unsigned r0 = tid / LPS, b0 = (tid % LPS) * BPL;
    const unsigned strided0 = r0 * (unsigned)pitch + (unsigned)(cg * SEG) + b0;
    const unsigned contig0 = (unsigned)(cg * (long long)SEG * ROWS) + r0 * SEG + b0;
    vec v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const unsigned so = OUT_STRIDED ? (unsigned)(i * RPI * SEG) : (unsigned)(i * RPI) * (unsigned)pitch;
      if constexpr (BPL == 16) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, OUT_STRIDED ? contig0 : strided0, so, 2);
      else v[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, OUT_STRIDED ? contig0 : strided0, so, 2);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const unsigned so = OUT_STRIDED ? (unsigned)(i * RPI) * (unsigned)pitch : (unsigned)(i * RPI * SEG);
      if constexpr (BPL == 16) __builtin_amdgcn_raw_buffer_store_b128(v[i], rd, OUT_STRIDED ? strided0 : contig0, so, 2);
      else __builtin_amdgcn_raw_buffer_store_b64(v[i], rd, OUT_STRIDED ? strided0 : contig0, so, 2);
    }
  }
}
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
template <int SEG, int ROWS, int WG, int BPL, int MAP, bool OS>
double run(char* in, char* out, size_t bytes, long long pitch, int gdiv) {
  const long long groups = (long long)(bytes / ((size_t)SEG * ROWS));
  long long grid = std::max(16ll, (groups / gdiv) / 16 * 16);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> t;
  for (int r = 0; r < 6; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((copy_cols<SEG, ROWS, WG, BPL, MAP, OS>), dim3((unsigned)grid), dim3(WG), 0, 0, in, out, groups, pitch);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) t.push_back(ms);
  }
  CK(hipGetLastError());
  return 2.0 * bytes / median(t) * 1e-9;
}
template <int SEG, int ROWS, int WG, int BPL>
void family(const char* name, char* in, char* out, size_t bytes, long long pitch) {
  printf("%-58s", name);
  for (int gdiv : {1, 4}) {
    printf(" | grid=groups/%d: in-strided %.2f / %.2f / %.2f  out-strided %.2f / %.2f / %.2f", gdiv,
           run<SEG, ROWS, WG, BPL, 0, false>(in, out, bytes, pitch, gdiv), run<SEG, ROWS, WG, BPL, 1, false>(in, out, bytes, pitch, gdiv),
           run<SEG, ROWS, WG, BPL, 2, false>(in, out, bytes, pitch, gdiv), run<SEG, ROWS, WG, BPL, 0, true>(in, out, bytes, pitch, gdiv),
           run<SEG, ROWS, WG, BPL, 1, true>(in, out, bytes, pitch, gdiv), run<SEG, ROWS, WG, BPL, 2, true>(in, out, bytes, pitch, gdiv));
  }
  printf("  TB/s (grid-stride / paired on one XCD / both halves in one block)\n");
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t bytes = (size_t)2 << 30;
  char *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
  fill_random<<<4096, 256>>>((unsigned*)in, bytes / 4); CK(hipDeviceSynchronize());
  // fp32 n = 2048 stage: 8 columns x 8 B = 64 B x 2048 rows at a 16 KiB pitch, 1024 lanes
  family<64, 2048, 1024, 8>("64 B x 2048 rows @ 16 KiB (fp32 2^22 stage, 8 B/lane)", in, out, bytes, 16384);
  family<128, 1024, 1024, 8>("128 B x 1024 rows @ 8 KiB (fp32 2^20 stage, 8 B/lane)", in, out, bytes, 8192);
  family<64, 1024, 512, 16>("64 B x 1024 rows @ 16 KiB (16 B/lane)", in, out, bytes, 16384);
  family<128, 1024, 512, 16>("128 B x 1024 rows @ 16 KiB (fp64 2^20 stage, 16 B/lane)", in, out, bytes, 16384);
  family<32, 1024, 256, 8>("32 B x 1024 rows @ 8 KiB", in, out, bytes, 8192);
  return 0;
}
