// Yardstick for UNPACKED "rows of a padded matrix" layouts of short transforms: copy rows of ROW bytes that sit at a pitch
// of PITCH bytes (only the ROW bytes of every pitch are read and written), lanes element-fastest at BPL bytes per lane --
// the access shape of the STAGED unpacked kernels (fp32 N = 16, ld = 20: 128-byte rows at a 160-byte pitch).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/copy_padded_rows.hip -o build/copy_padded_rows
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

template <int BPL, int WG, int PER_LANE>
__global__ __launch_bounds__(WG) void copy_rows(const char* __restrict__ in, char* __restrict__ out, long long rows, int row_bytes, int pitch) {
  using vec = typename lane_vec<BPL>::type;
  const int lanes_per_row = row_bytes / BPL;
  const long long pieces = rows * lanes_per_row;  // BPL-byte pieces of payload
  for (long long base = (long long)blockIdx.x * WG * PER_LANE; base < pieces; base += (long long)gridDim.x * WG * PER_LANE) {
    vec v[PER_LANE];
    long long off[PER_LANE];
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
      const long long p = base + (long long)i * WG + threadIdx.x;
      const long long row = p / lanes_per_row, col = p % lanes_per_row;
      off[i] = p < pieces ? row * pitch + col * BPL : -1;
      if (off[i] >= 0) v[i] = __builtin_nontemporal_load(reinterpret_cast<const vec*>(in + off[i]));
    }
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i)
      if (off[i] >= 0) __builtin_nontemporal_store(v[i], reinterpret_cast<vec*>(out + off[i]));
  }
}

template <int BPL>
void run(char* in, char* out, size_t bytes, int row_bytes, int pitch, int cus) {
  const long long rows = (long long)(bytes / pitch);
  const double payload = 2.0 * rows * row_bytes;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("rows of %4d B at a pitch of %4d B, %2d B per lane:", row_bytes, pitch, BPL);
  for (int mult : {4, 8, 16}) {
    std::vector<float> t;
    for (int r = 0; r < 6; ++r) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((copy_rows<BPL, 256, 8>), dim3(mult * cus), dim3(256), 0, 0, in, out, rows, row_bytes, pitch);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("  x%-2d %.2f TB/s (%.3f of 8)", mult, payload / t[t.size() / 2] * 1e-9, payload / t[t.size() / 2] * 1e-9 / 8);
  }
  printf("   payload bytes\n");
  CK(hipGetLastError());
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  char *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); const int cus = prop.multiProcessorCount;
  run<8>(in, out, bytes, 128, 128, cus);    // fp32 N = 16 packed
  run<8>(in, out, bytes, 128, 160, cus);    // fp32 N = 16, ld = 20
  run<16>(in, out, bytes, 256, 320, cus);   // fp64 N = 16, ld = 20
  run<8>(in, out, bytes, 512, 640, cus);    // fp32 N = 64, ld = 80
  run<16>(in, out, bytes, 1024, 1280, cus); // fp64 N = 64, ld = 80
  run<8>(in, out, bytes, 128, 256, cus);    // every second 128-byte line
  return 0;
}
