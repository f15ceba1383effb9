// Infinity-Cache yardsticks for the two-launch plans.
//
// Question: a two-launch plan (four-step stages A/B, 2-D passes 1/2) runs chunk by chunk with a chunk's intermediate
// sized to the 256 MiB Infinity Cache.  Its stage kernels move 5.2-6.3 TB/s (read + write).  Is that the on-die
// path's limit or the kernels'?  This probe runs plain copies (no arithmetic, no LDS) with the stage kernels'
// segment shapes and cache policies:
//   W  writer: IN chunk (HBM, nt loads)            -> MID (default / sc1 / nt stores)
//   R  reader: MID (default / nt loads)            -> OUT chunk (HBM, nt stores)
//   C  one-pass copy IN chunk -> OUT chunk, nt / nt  (the "HBM pass" rate of the same shape)
//   F  fused launch: even work-groups run W of chunk c + 1 (into MID[(c+1) & 1]), odd ones R of chunk c
//   RO / WO  read-only / write-only sweeps of a buffer of S bytes, repeated (Infinity-Cache read / write ceilings)
// over intermediate sizes S = 32 ... 512 MiB, random data (the on-die path is data dependent, profiles/r2_notes.md).
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/ic_yardstick.hip -o build/ic_yardstick
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr int AUX_DEF = 0, AUX_NT = 2, AUX_SC1 = 0x10;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

__global__ void fill_random(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + seed * 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
    // a float in [-1, 1): random mantissa, exponent of 1.0 -> [1, 2) - 1.5 ... keep it simple: random mantissa bits
    unsigned m = (unsigned)(z >> 40) | 0x3F800000u;  // [1, 2)
    float f = (__uint_as_float(m) - 1.5f) * 2.0f;
    p[i] = __float_as_uint(f);
  }
}

// One "group" = ROWS segments of SEG bytes = what one stage work-group holds (128 KiB for the C3 stages).
// SHAPE 0: contiguous in, contiguous out; 1: strided in (pitch), contiguous (group-major) out  [stage A];
//       2: contiguous (group-major) in, strided out [stage B].
// The group's IT x 16 B per lane are all loaded before the first store (as the FFT kernels do).
template <int SEG, int ROWS, int WG, int SHAPE>
struct shape_t {
  static constexpr int LPS = SEG / 16;
  static constexpr int RPI = WG / LPS;
  static constexpr int IT = ROWS / RPI;
  static constexpr long long GROUP_BYTES = (long long)SEG * ROWS;
};

template <int SEG, int ROWS, int WG, int SHAPE, int LD_AUX, int ST_AUX>
__device__ __forceinline__ void copy_group(const char* src, char* dst, long long g, long long pitch, unsigned chunk_bytes) {
  using S = shape_t<SEG, ROWS, WG, SHAPE>;
  const int tid = threadIdx.x;
  const long long gpi = pitch / SEG;  // groups per image (image = ROWS x pitch bytes)
  const long long img = g / gpi, cg = g % gpi;
  const long long img_bytes = (long long)ROWS * pitch;
  const unsigned r0 = tid / S::LPS, b0 = (tid % S::LPS) * 16;
  const unsigned contig0 = (unsigned)(cg * S::GROUP_BYTES) + r0 * SEG + b0;
  const unsigned strided0 = r0 * (unsigned)pitch + (unsigned)(cg * SEG) + b0;
  auto rs = rsrc_of(src + img * img_bytes, (unsigned)img_bytes);
  auto rd = rsrc_of(dst + img * img_bytes, (unsigned)img_bytes);
  v4u v[S::IT];
#pragma unroll
  for (int i = 0; i < S::IT; ++i) {
    const unsigned step = (SHAPE == 1) ? (unsigned)(i * S::RPI) * (unsigned)pitch : (unsigned)(i * S::RPI * SEG);
    v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (SHAPE == 1) ? strided0 : contig0, step, LD_AUX);
  }
#pragma unroll
  for (int i = 0; i < S::IT; ++i) {
    const unsigned step = (SHAPE == 2) ? (unsigned)(i * S::RPI) * (unsigned)pitch : (unsigned)(i * S::RPI * SEG);
    __builtin_amdgcn_raw_buffer_store_b128(v[i], rd, (SHAPE == 2) ? strided0 : contig0, step, ST_AUX);
  }
}

template <int SEG, int ROWS, int WG, int SHAPE, int LD_AUX, int ST_AUX>
__global__ __launch_bounds__(WG) void copy_kernel(const char* src, char* dst, long long groups, long long pitch) {
  for (long long g = blockIdx.x; g < groups; g += gridDim.x)
    copy_group<SEG, ROWS, WG, SHAPE, LD_AUX, ST_AUX>(src, dst, g, pitch, 0);
}

// fused: even work-groups write (shape SW, nt loads, WST stores) src_w -> mid_w; odd ones read (shape SR) mid_r -> dst_r
template <int SEG, int ROWS, int WG, int SW, int SR, int WST, int RLD>
__global__ __launch_bounds__(WG) void fused_kernel(const char* src_w, char* mid_w, const char* mid_r, char* dst_r,
                                                  long long groups_w, long long groups_r, long long pitch) {
  const unsigned role = blockIdx.x & 1, idx = blockIdx.x >> 1, n = gridDim.x >> 1;
  if (role == 0) {
    for (long long g = idx; g < groups_w; g += n) copy_group<SEG, ROWS, WG, SW, AUX_NT, WST>(src_w, mid_w, g, pitch, 0);
  } else {
    for (long long g = idx; g < groups_r; g += n) copy_group<SEG, ROWS, WG, SR, RLD, AUX_NT>(mid_r, dst_r, g, pitch, 0);
  }
}

template <int LD_AUX>
__global__ __launch_bounds__(256) void read_only(const char* src, unsigned* sink, long long bytes) {
  const long long tiles = bytes / 4096;  // 256 lanes x 16 B
  v4u acc = {0, 0, 0, 0};
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    auto rs = rsrc_of(src + t * 4096, 4096);
    v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16, 0, LD_AUX);
    acc ^= v;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
template <int ST_AUX>
__global__ __launch_bounds__(256) void write_only(char* dst, long long bytes, unsigned seed) {
  const long long tiles = bytes / 4096;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    auto rd = rsrc_of(dst + t * 4096, 4096);
    unsigned h = (unsigned)t * 2654435761u + threadIdx.x * 40503u + seed;
    v4u v = {h, h * 3u + 1u, h ^ 0x5bd1e995u, h * 7u};
    __builtin_amdgcn_raw_buffer_store_b128(v, rd, threadIdx.x * 16, 0, ST_AUX);
  }
}

struct timer {
  hipEvent_t a, b;
  timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  void start() { CK(hipEventRecord(a)); }
  float stop() { CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
};
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

static int g_cus = 256;
static const size_t TOTAL = (size_t)2 << 30;

template <int SEG, int ROWS, int WG, int SHAPE, int LD, int ST>
void launch_copy(const char* s, char* d, size_t bytes, long long pitch, int gdiv) {
  const long long groups = (long long)(bytes / ((size_t)SEG * ROWS));
  long long grid = std::max(1ll, groups / gdiv);
  hipLaunchKernelGGL((copy_kernel<SEG, ROWS, WG, SHAPE, LD, ST>), dim3((unsigned)grid), dim3(WG), 0, 0, s, d, groups, pitch);
}

// one full pipeline over TOTAL bytes in chunks of S: per-kernel event times (writer, reader) and the wall total
template <int SEG, int ROWS, int WG, int SW, int SR, int WST, int RLD>
void pipeline(const char* name, char* in, char* mid, char* out, size_t S, long long pitch, int gdiv) {
  const int nchunks = (int)(TOTAL / S);
  std::vector<hipEvent_t> ev(2 * nchunks + 1);
  for (auto& e : ev) CK(hipEventCreate(&e));
  std::vector<double> tw, tr, tt, tplain;
  timer T;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(ev[0]));
    for (int c = 0; c < nchunks; ++c) {
      launch_copy<SEG, ROWS, WG, SW, AUX_NT, WST>(in + c * S, mid, S, pitch, gdiv);
      CK(hipEventRecord(ev[2 * c + 1]));
      launch_copy<SEG, ROWS, WG, SR, RLD, AUX_NT>(mid, out + c * S, S, pitch, gdiv);
      CK(hipEventRecord(ev[2 * c + 2]));
    }
    CK(hipEventSynchronize(ev[2 * nchunks]));
    double w = 0, r = 0;
    for (int c = 0; c < nchunks; ++c) {
      float ms; CK(hipEventElapsedTime(&ms, ev[2 * c], ev[2 * c + 1])); w += ms;
      CK(hipEventElapsedTime(&ms, ev[2 * c + 1], ev[2 * c + 2])); r += ms;
    }
    // the same sequence without events in between (what a plan's execute issues)
    T.start();
    for (int c = 0; c < nchunks; ++c) {
      launch_copy<SEG, ROWS, WG, SW, AUX_NT, WST>(in + c * S, mid, S, pitch, gdiv);
      launch_copy<SEG, ROWS, WG, SR, RLD, AUX_NT>(mid, out + c * S, S, pitch, gdiv);
    }
    double plain = T.stop();
    if (rep) { tw.push_back(w); tr.push_back(r); tt.push_back(w + r); tplain.push_back(plain); }
  }
  const double gb = 2.0 * TOTAL * 1e-9;  // bytes moved per kernel type over the whole sequence (read + write)
  printf("%-44s S=%4zu MiB  writer %.2f  reader %.2f  TB/s | pair %.3f ms (events) %.3f ms (plain) = %.3f of 8 TB/s on 1x bytes\n", name, S >> 20,
         gb / median(tw), gb / median(tr), median(tt), median(tplain), gb / median(tplain) / 8.0);
  for (auto& e : ev) CK(hipEventDestroy(e));
}

template <int SEG, int ROWS, int WG, int SW, int SR, int WST, int RLD>
void fused(const char* name, char* in, char* mid, char* out, size_t S, long long pitch, int gdiv) {
  // MID holds two buffers of S bytes; launch c runs W(c) [c < n] beside R(c - 1) [c > 0]
  const int nchunks = (int)(TOTAL / S);
  const long long groups = (long long)(S / ((size_t)SEG * ROWS));
  std::vector<double> t;
  timer T;
  for (int rep = 0; rep < 5; ++rep) {
    T.start();
    for (int c = 0; c <= nchunks; ++c) {
      const long long gw = c < nchunks ? groups : 0, gr = c > 0 ? groups : 0;
      long long grid = 2 * std::max(1ll, groups / gdiv);
      hipLaunchKernelGGL((fused_kernel<SEG, ROWS, WG, SW, SR, WST, RLD>), dim3((unsigned)grid), dim3(WG), 0, 0,
                         in + (size_t)(c % nchunks) * S, mid + (size_t)(c & 1) * S, mid + (size_t)((c + 1) & 1) * S,
                         out + (size_t)((c + nchunks - 1) % nchunks) * S, gw, gr, pitch);
    }
    double ms = T.stop();
    if (rep) t.push_back(ms);
  }
  const double gb = 2.0 * TOTAL * 1e-9;
  printf("%-44s S=%4zu MiB x2 fused W(c+1)|R(c): total %.3f ms = %.3f of 8 TB/s on 1x bytes (%d launches)\n", name, S >> 20, median(t),
         gb / median(t) / 8.0, nchunks + 1);
}

template <int SEG, int ROWS, int WG, int SHAPE>
void onepass(const char* name, char* in, char* out, long long pitch, int gdiv) {
  timer T; std::vector<double> t;
  for (int rep = 0; rep < 6; ++rep) {
    T.start();
    launch_copy<SEG, ROWS, WG, SHAPE, AUX_NT, AUX_NT>(in, out, TOTAL, pitch, gdiv);
    double ms = T.stop(); if (rep) t.push_back(ms);
  }
  printf("%-44s one pass HBM->HBM nt/nt: %.2f TB/s\n", name, 2.0 * TOTAL * 1e-9 / median(t));
}

template <int SEG, int ROWS, int WG, int SW, int SR>
void family(const char* name, char* in, char* mid, char* out, long long pitch, int gdiv) {
  char buf[128];
  printf("---- %s: segment %d B, %d rows, %d lanes, pitch %lld, grid = groups/%d\n", name, SEG, ROWS, WG, pitch, gdiv);
  snprintf(buf, sizeof buf, "%s writer-shape", name); onepass<SEG, ROWS, WG, SW>(buf, in, out, pitch, gdiv);
  snprintf(buf, sizeof buf, "%s reader-shape", name); onepass<SEG, ROWS, WG, SR>(buf, in, out, pitch, gdiv);
  for (size_t S : {(size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)192 << 20, (size_t)256 << 20, (size_t)512 << 20}) {
    if (TOTAL % S) continue;
    snprintf(buf, sizeof buf, "%s W st=sc1 / R ld=default", name);
    pipeline<SEG, ROWS, WG, SW, SR, AUX_SC1, AUX_DEF>(buf, in, mid, out, S, pitch, gdiv);
    if (S == ((size_t)128 << 20) || S == ((size_t)256 << 20)) {
      snprintf(buf, sizeof buf, "%s W st=default / R ld=default", name);
      pipeline<SEG, ROWS, WG, SW, SR, AUX_DEF, AUX_DEF>(buf, in, mid, out, S, pitch, gdiv);
      snprintf(buf, sizeof buf, "%s W st=nt / R ld=nt (no cache)", name);
      pipeline<SEG, ROWS, WG, SW, SR, AUX_NT, AUX_NT>(buf, in, mid, out, S, pitch, gdiv);
      snprintf(buf, sizeof buf, "%s W st=sc1 / R ld=sc1", name);
      pipeline<SEG, ROWS, WG, SW, SR, AUX_SC1, AUX_SC1>(buf, in, mid, out, S, pitch, gdiv);
    }
  }
  for (size_t S : {(size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)256 << 20}) {
    snprintf(buf, sizeof buf, "%s", name);
    fused<SEG, ROWS, WG, SW, SR, AUX_SC1, AUX_DEF>(buf, in, mid, out, S, pitch, gdiv);
  }
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); g_cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, L2 %d MiB\n", prop.name, g_cus, prop.l2CacheSize >> 20);
  char *in, *mid, *out; unsigned* sink;
  CK(hipMalloc(&in, TOTAL)); CK(hipMalloc(&out, TOTAL)); CK(hipMalloc(&mid, (size_t)1 << 30)); CK(hipMalloc(&sink, 64));
  fill_random<<<4096, 256>>>((unsigned*)in, TOTAL / 4, 1);
  fill_random<<<4096, 256>>>((unsigned*)out, TOTAL / 4, 2);
  fill_random<<<4096, 256>>>((unsigned*)mid, ((size_t)1 << 30) / 4, 3);
  CK(hipDeviceSynchronize());
  const char* only = argc > 1 ? argv[1] : "all";
  auto want = [&](const char* k) { return !strcmp(only, "all") || !strcmp(only, k); };

  if (want("sweep")) {
    printf("---- read-only / write-only sweeps of S bytes, 10 passes each after one warm-up pass (TB/s)\n");
    timer T;
    for (size_t mib : {16, 32, 64, 128, 192, 224, 256, 320, 384, 512, 1024}) {
      const long long S = (long long)mib << 20;
      double r[2], w[3];
      for (int k = 0; k < 2; ++k) {
        std::vector<double> t;
        for (int rep = 0; rep < 4; ++rep) {
          T.start();
          for (int p = 0; p < 10; ++p) {
            if (k == 0) hipLaunchKernelGGL(read_only<AUX_DEF>, dim3(g_cus * 16), dim3(256), 0, 0, mid, sink, S);
            else hipLaunchKernelGGL(read_only<AUX_NT>, dim3(g_cus * 16), dim3(256), 0, 0, mid, sink, S);
          }
          double ms = T.stop(); if (rep) t.push_back(ms);
        }
        r[k] = 10.0 * S * 1e-9 / median(t);
      }
      for (int k = 0; k < 3; ++k) {
        std::vector<double> t;
        for (int rep = 0; rep < 4; ++rep) {
          T.start();
          for (int p = 0; p < 10; ++p) {
            if (k == 0) hipLaunchKernelGGL(write_only<AUX_DEF>, dim3(g_cus * 16), dim3(256), 0, 0, mid, S, (unsigned)p);
            else if (k == 1) hipLaunchKernelGGL(write_only<AUX_SC1>, dim3(g_cus * 16), dim3(256), 0, 0, mid, S, (unsigned)p);
            else hipLaunchKernelGGL(write_only<AUX_NT>, dim3(g_cus * 16), dim3(256), 0, 0, mid, S, (unsigned)p);
          }
          double ms = T.stop(); if (rep) t.push_back(ms);
        }
        w[k] = 10.0 * S * 1e-9 / median(t);
      }
      printf("S=%5zu MiB  read default %.2f  nt %.2f | write default %.2f  sc1 %.2f  nt %.2f\n", mib, r[0], r[1], w[0], w[1], w[2]);
    }
    fill_random<<<4096, 256>>>((unsigned*)mid, ((size_t)1 << 30) / 4, 3);
    CK(hipDeviceSynchronize());
  }
  // C3 stage shapes: fp64 N = 2^20 as 1024 x 1024, 8 columns (128 B) x 1024 rows per work-group of 512 lanes, pitch 16 KiB,
  // four groups per work-group
  if (want("c3")) family<128, 1024, 512, 1, 2>("C3-shape (A: strided->tiles, B: tiles->strided)", in, mid, out, 16384, 4);
  // contiguous both sides, same group size (the best case for the on-die path)
  if (want("contig")) family<128, 1024, 512, 0, 0>("contiguous 128 KiB groups", in, mid, out, 16384, 4);
  // ref65536 / C5-like: 256-byte and 512-byte segments, 256 rows
  if (want("seg256")) family<256, 256, 512, 1, 2>("256 B segments x 256 rows (fp32 65536)", in, mid, out, 2048, 4);
  if (want("seg512")) family<512, 128, 256, 0, 2>("C5-shape (rows -> 512 B column segments x 128)", in, mid, out, 65536, 2);
  // small work-groups at high occupancy (4 KiB granules: the best copy shape seen in round 1)
  if (want("small")) family<256, 16, 256, 0, 0>("4 KiB granules, 256 lanes", in, mid, out, 4096, 16);
  CK(hipGetLastError());
  return 0;
}
