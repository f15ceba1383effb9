// Yardstick for the column passes: copy a [images][1024 rows][row_bytes] array where every work-group moves a
// "column group" = 1024 segments of SEG bytes at a stride of row_bytes (the access shape of the strided FFT kernels).
// Variants: strided -> strided (stage A/B without row staging), strided -> contiguous, contiguous -> strided.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <int BPL> struct lane_vec { using type = v4f; };
template <> struct lane_vec<8> { using type = v2f; };

// MODE 0: strided in, strided out; 1: strided in, contiguous out; 2: contiguous in, strided out
// BPL: bytes per lane (16: the fp64 kernels' element, 8: the fp32 kernels')
template <int SEG, int WG, int MODE, int ROWS, int BPL = 16>
__global__ __launch_bounds__(WG) void copy_cols(const char* __restrict__ in, char* __restrict__ out, long long groups,
                                                long long row_bytes) {
  using vec = typename lane_vec<BPL>::type;
  constexpr int LPS = SEG / BPL;           // lanes per segment
  constexpr int RPI = WG / LPS;            // rows per iteration
  constexpr int IT = ROWS / RPI;
  const int tid = threadIdx.x;
  const long long gpi = row_bytes / SEG;   // groups per image
  for (long long g = blockIdx.x; g < groups; g += gridDim.x) {
    const long long img = g / gpi, cg = g % gpi;
    const char* src = in + img * ROWS * row_bytes;
    char* dst = out + img * ROWS * row_bytes;
    vec v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const long long r = tid / LPS + (long long)i * RPI;
      const long long so = (MODE == 2) ? (cg * ROWS * SEG + r * SEG + (tid % LPS) * BPL) : (r * row_bytes + cg * SEG + (tid % LPS) * BPL);
      v[i] = __builtin_nontemporal_load(reinterpret_cast<const vec*>(src + so));
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const long long r = tid / LPS + (long long)i * RPI;
      const long long d_ = (MODE == 1) ? (cg * ROWS * SEG + r * SEG + (tid % LPS) * BPL) : (r * row_bytes + cg * SEG + (tid % LPS) * BPL);
      __builtin_nontemporal_store(v[i], reinterpret_cast<vec*>(dst + d_));
    }
  }
}

template <int SEG, int WG, int MODE, int ROWS, int BPL = 16>
void run(const char* name, char* in, char* out, size_t bytes, long long row_bytes, int cus) {
  const long long groups = (long long)(bytes / ((size_t)ROWS * SEG));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-34s seg=%4d wg=%4d rowB=%6lld", name, SEG, WG, row_bytes);
  for (int mult : {1, 2, 4, 8, 16, 0}) {
    long long grid = mult ? (long long)mult * cus : groups;
    grid = std::min(grid, groups);
    std::vector<float> t;
    for (int r = 0; r < 6; ++r) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((copy_cols<SEG, WG, MODE, ROWS, BPL>), dim3((unsigned)grid), dim3(WG), 0, 0, in, out, groups, row_bytes);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("  x%-2d %.2f", mult, 2.0 * bytes / t[t.size() / 2] * 1e-9);
  }
  printf("  TB/s\n");
  CK(hipGetLastError());
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  char *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMemset(in, 1, bytes));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); const int cus = prop.multiProcessorCount;
  // C3 shape: rows of 1024 complex128 = 16 KiB; C5 shape: rows of 1024 complex64 = 8 KiB
  if (getenv("COPY_BPL") != nullptr) {  // 8 against 16 bytes per lane at the same segment sizes (fp32 vs fp64 column kernels)
    for (long long rb : {8192ll}) {
      run<128, 512, 0, 1024, 16>("strided->strided 16 B/lane", in, out, bytes, rb, cus);
      run<128, 512, 0, 1024, 8>("strided->strided  8 B/lane", in, out, bytes, rb, cus);
      run<128, 1024, 0, 1024, 8>("strided->strided  8 B/lane", in, out, bytes, rb, cus);
      run<128, 512, 1, 1024, 16>("strided->contiguous 16 B/lane", in, out, bytes, rb, cus);
      run<128, 512, 1, 1024, 8>("strided->contiguous  8 B/lane", in, out, bytes, rb, cus);
      run<128, 1024, 1, 1024, 8>("strided->contiguous  8 B/lane", in, out, bytes, rb, cus);
      run<128, 512, 2, 1024, 16>("contiguous->strided 16 B/lane", in, out, bytes, rb, cus);
      run<128, 512, 2, 1024, 8>("contiguous->strided  8 B/lane", in, out, bytes, rb, cus);
      run<256, 512, 0, 1024, 16>("strided->strided 16 B/lane", in, out, bytes, rb, cus);
      run<256, 512, 0, 1024, 8>("strided->strided  8 B/lane", in, out, bytes, rb, cus);
      run<256, 1024, 0, 1024, 8>("strided->strided  8 B/lane", in, out, bytes, rb, cus);
    }
    return 0;
  }
  if (getenv("COPY_PITCH") != nullptr) {  // BATCH_INTERLEAVED N = 1024: 128-byte segments, 1024 rows, the row pitch = batch x 8 B
    for (long long rb : {65536ll, 262144ll, 1048576ll, 2097152ll}) {
      run<128, 1024, 0, 1024, 8>("BI N=1024 shape, 8 B/lane", in, out, bytes, rb, cus);
      run<128, 512, 0, 1024, 16>("BI N=1024 shape, 16 B/lane", in, out, bytes, rb, cus);
      run<256, 1024, 0, 1024, 8>("32 columns, 8 B/lane", in, out, bytes, rb, cus);
    }
    return 0;
  }
  for (long long rb : {16384ll, 8192ll}) {
    run<64, 256, 0, 1024>("strided->strided", in, out, bytes, rb, cus);
    run<128, 512, 0, 1024>("strided->strided", in, out, bytes, rb, cus);
    run<256, 512, 0, 1024>("strided->strided", in, out, bytes, rb, cus);
    run<512, 1024, 0, 1024>("strided->strided", in, out, bytes, rb, cus);
    run<128, 512, 1, 1024>("strided->contiguous", in, out, bytes, rb, cus);
    run<256, 512, 1, 1024>("strided->contiguous", in, out, bytes, rb, cus);
    run<128, 512, 2, 1024>("contiguous->strided", in, out, bytes, rb, cus);
    run<256, 512, 2, 1024>("contiguous->strided", in, out, bytes, rb, cus);
  }
  return 0;
}
