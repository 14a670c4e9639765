// tap_paths.hip — what does it cost to bring the 16 taps of every pixel of a MINIFYING bicubic pass into registers when
// the bytes are already on chip (L2 / Infinity Cache resident source)?
//
// row_gather.hip measures how fast the windows of BASELINE configs[3] arrive from HBM; the ablations of the kernel
// (profiles/r04_rect_eqr_ablations.txt) say the 16 gathers per pixel cost the same whether they hit in cache or not.
// This program times that tap path alone, per 16 x 4 pass (64 pixels, pixels SX texels apart in x, rows SY apart):
//
//   mode 0  lane = pixel: 4 tap rows x 4 consecutive texels as 16 dwordx4 gathers (the kernel's direct path);
//   mode 1  lane = (tap column i = lane / 16, pixel column p = lane % 16): for each of the 4 pixel rows of the pass and each
//           tap row one dwordx4 — the same 16 instructions and the same bytes, but a wave instruction now touches 16
//           segments of 64 contiguous bytes instead of 64 texels of 64 different segments;
//   mode 2  the pass window by LDS-DMA (rows of the window, 1 KiB per instruction), then 16 ds_read_b128 per lane.
//
// Output: microseconds for a 4096^2-equivalent number of passes and nanoseconds per pass and CU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/tap_paths.hip -o tools/microbench/tap_paths
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); \
      return 1;                                                               \
    }                                                                         \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

struct Args {
  const char *src;    // RGBA image, rows of row_bytes
  float *sink;
  unsigned row_bytes;
  int rows, cols;     // texels
  int sx16, sy16;     // pixel spacing in 1/16 texel
  int passes;         // passes per wavefront
  int mask;           // passes of the whole launch wrap inside (mask + 1) pass positions: the footprint
};

extern __shared__ char s_dyn[];

__device__ inline void pass_origin(const Args &A, int pass, int &x0, int &y0) {
  // pass positions tile the image: 16 pixels * sx wide, 4 rows * sy tall
  const int pw = (16 * A.sx16 + 15) >> 4, ph = (4 * A.sy16 + 15) >> 4;
  const int per_row = (A.cols - 8) / pw, n_rows = (A.rows - 8) / ph;
  const int q = pass & A.mask;
  x0 = (q % per_row) * pw;
  y0 = ((q / per_row) % n_rows) * ph;
}

__global__ __launch_bounds__(64) void k_pixel(const Args A) {
  const int lane = (int)threadIdx.x, p = lane & 15, j = lane >> 4;
  v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int w = 0; w < A.passes; ++w) {
    int x0, y0;
    pass_origin(A, (int)blockIdx.x * A.passes + w, x0, y0);
    const char *b = A.src + (size_t)(y0 + ((j * A.sy16) >> 4)) * A.row_bytes + (size_t)(x0 + ((p * A.sx16) >> 4)) * 16;
    v4f t[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) t[r][i] = *reinterpret_cast<const v4f *>(b + (size_t)r * A.row_bytes + 16 * i);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += t[r][i];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) A.sink[blockIdx.x] = acc.x;
}

__global__ __launch_bounds__(64) void k_columns(const Args A) {
  const int lane = (int)threadIdx.x, p = lane & 15, i = lane >> 4;
  v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int w = 0; w < A.passes; ++w) {
    int x0, y0;
    pass_origin(A, (int)blockIdx.x * A.passes + w, x0, y0);
    const char *b = A.src + (size_t)y0 * A.row_bytes + (size_t)(x0 + ((p * A.sx16) >> 4) + i) * 16;
    v4f t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) t[j][r] = *reinterpret_cast<const v4f *>(b + (size_t)(((j * A.sy16) >> 4) + r) * A.row_bytes);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc += t[j][r];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) A.sink[blockIdx.x] = acc.x;
}

__global__ __launch_bounds__(64) void k_window(const Args A) {
  const int lane = (int)threadIdx.x, p = lane & 15, j = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)s_dyn;
  const int bw = ((15 * A.sx16) >> 4) + 4, bh = ((3 * A.sy16) >> 4) + 4, pitch = bw | 1;
  const int chunks = (bw + 63) >> 6;
  v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
  const v4f *win = reinterpret_cast<const v4f *>(s_dyn);
  for (int w = 0; w < A.passes; ++w) {
    int x0, y0;
    pass_origin(A, (int)blockIdx.x * A.passes + w, x0, y0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int r = 0; r < bh; ++r)
      for (int c = 0; c < chunks; ++c) {
        const int col = c * 64 + lane;
        if (col < bw) {
          const char *row = A.src + (size_t)(y0 + r) * A.row_bytes + (size_t)x0 * 16;
          const unsigned lds = lds0 + (unsigned)((r * pitch + c * 64) * 16);
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"((unsigned)col * 16u), "s"(row) : "memory");
        }
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const v4f *t0 = win + ((j * A.sy16) >> 4) * pitch + ((p * A.sx16) >> 4);
    v4f t[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) t[r][i] = t0[r * pitch + i];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += t[r][i];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) A.sink[blockIdx.x] = acc.x;
}

int main(int argc, char **argv) {
  int W = 4096, reps = 5;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--size")) W = atoi(argv[i + 1]);
    if (!strcmp(argv[i], "--reps")) reps = atoi(argv[i + 1]);
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::printf("# %s, %d CUs; RGBA source %d x %d\n", prop.name, cus, W, W);
  char *src;
  CK(hipMalloc(&src, (size_t)W * W * 16 + 4096));
  CK(hipMemset(src, 1, (size_t)W * W * 16 + 4096));
  float *sink;
  CK(hipMalloc(&sink, sizeof(float) * (1 << 22)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // passes of a frame whose in-view part is 37 % of 4096^2 output pixels
  const int total_passes = (int)(0.37 * 4096.0 * 4096.0 / 64.0);
  auto run = [&](const char *label, int mode, int sx16, int sy16, int waves_per_cu, int footprint_passes) -> int {
    Args A{};
    A.src = src;
    A.sink = sink;
    A.row_bytes = (unsigned)W * 16u;
    A.rows = A.cols = W;
    A.sx16 = sx16;
    A.sy16 = sy16;
    A.passes = 8;
    A.mask = footprint_passes - 1;
    const unsigned grid = (unsigned)(total_passes / A.passes);
    size_t lds = (size_t)(160 * 1024) / (size_t)waves_per_cu;
    lds = lds / 1024 * 1024;
    if (lds > 64 * 1024) lds = 64 * 1024;
    const int bw = ((15 * sx16) >> 4) + 4, bh = ((3 * sy16) >> 4) + 4;
    if (mode == 2 && (size_t)((bw | 1) * bh * 16) > lds) {
      std::printf("%-22s window of %d x %d texels does not fit %zu bytes at %d waves/CU\n", label, bw, bh, lds, waves_per_cu);
      return 0;
    }
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k_pixel, dim3(grid), dim3(64), lds, 0, A);
      else if (mode == 1) hipLaunchKernelGGL(k_columns, dim3(grid), dim3(64), lds, 0, A);
      else hipLaunchKernelGGL(k_window, dim3(grid), dim3(64), lds, 0, A);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms < best) best = ms;
    }
    const double passes = (double)grid * A.passes;
    std::printf("%-22s sx %.2f sy %.2f  %2d waves/CU  footprint %7d passes  %7.1f us  %6.1f ns per pass and CU  (%.0f GB/s of taps)\n", label, sx16 / 16.0,
                sy16 / 16.0, waves_per_cu, footprint_passes, best * 1e3, best * 1e6 / passes * cus, passes * 16384.0 / (best * 1e-3) / 1e9);
    return 0;
  };
  const char *names[3] = {"lane = pixel", "lane = (column, pixel)", "LDS window"};
  for (int foot : {1 << 10, 1 << 14, 1 << 20})
    for (int s : {0, 1, 2}) {
      const int sx16 = s == 0 ? 54 : s == 1 ? 72 : 16, sy16 = s == 0 ? 52 : s == 1 ? 40 : 16; // 3.4 x 3.25 (the median in-view pass of rect -> equirect at 4K), 4.5 x 2.5, 1 x 1
      for (int mode = 0; mode < 3; ++mode)
        for (int waves : {8, 16, 32}) {
          if (mode == 2 && waves == 32) continue;
          if (mode == 2 && waves == 16 && s != 2) continue;
          if (run(names[mode], mode, sx16, sy16, waves, foot)) return 1;
        }
    }
  return 0;
}
