// row_gather.hip — how fast can one MI355X fetch the source windows of a MINIFYING bicubic mapping?
//
// BASELINE configs[3] (rectilinear -> equirectangular, RGBAZ) spends its time in blocks whose pixels are 3-5 source
// texels apart in x and 1.5-3 in y: a 16 x 4 pass reads ~11 consecutive source rows of ~1340 contiguous bytes, the
// next pass starts 8 rows further down (3 rows of halo), the next block sits 1280 bytes to the right.  The kernel's
// counters say that phase runs at the rate at which an L1 keeps misses in flight.  This
// program measures that rate directly, without any arithmetic, for the ways such rows can be requested:
//
//   mode 0  rows as coalesced per-lane loads into registers (global_load_dwordx4, lane i = bytes 16 i .. 16 i + 15 of
//           the row), DEPTH rows requested before the first is consumed;
//   mode 1  rows by LDS-DMA (global_load_lds_dwordx4, no VGPRs), a whole window requested, then vmcnt(0);
//   mode 2  the direct path of the kernels: lane = output pixel, 4 tap rows x 5 dwordx4 per lane at an 80-byte pixel
//           stride (20-byte texels, 4 texels per pixel), rows of a pass 2 source rows apart.
//
// Swept: wavefronts per CU (limited through the LDS a workgroup asks for), rows in flight per wavefront, row length,
// 64- vs 128-byte alignment of the row starts, texel size.  Sources rotate over more memory than the Infinity Cache.
// Output: useful GB/s (bytes of the windows, halos counted once per window) and the requests offered per CU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/row_gather.hip -o tools/microbench/row_gather
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__);     \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

struct Args {
  const char *src;     // image: rows of row_bytes
  float *sink;         // one float per wavefront (keeps the loads alive)
  unsigned row_bytes;  // bytes per source row
  int rows;            // source rows
  int win_rows;        // R: consecutive source rows per window
  int win_bytes;       // L: contiguous bytes per window row
  int advance;         // source rows from one window of a wavefront to its next
  int windows;         // windows per wavefront (a strip walking down)
  int tiles_x;         // windows side by side
  int align_off;       // byte offset added to every row start (0: 128-byte aligned rows when win_bytes % 128 == 0)
};

extern __shared__ char s_dyn[];

// mode 0: coalesced rows into registers, Depth rows in flight
template <int Depth> __global__ __launch_bounds__(64) void rows_to_regs(const Args A) {
  const int lane = (int)threadIdx.x;
  const int tx = (int)blockIdx.x % A.tiles_x, ty = (int)blockIdx.x / A.tiles_x;
  const int chunks = (A.win_bytes + 1023) / 1024; // dwordx4 instructions per row
  v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int w = 0; w < A.windows; ++w) {
    const int y0 = (ty * A.windows + w) * A.advance;
    if (y0 + A.win_rows > A.rows) break;
    const char *base = A.src + (size_t)y0 * A.row_bytes + (size_t)tx * A.win_bytes + A.align_off;
    for (int r0 = 0; r0 < A.win_rows; r0 += Depth) {
      v4f v[Depth];
#pragma unroll
      for (int d = 0; d < Depth; ++d) {
        v[d] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
        if (r0 + d < A.win_rows)
          for (int c = 0; c < chunks; ++c) {
            const int off = c * 1024 + lane * 16;
            if (off < A.win_bytes) v[d] += __builtin_nontemporal_load(reinterpret_cast<const v4f *>(base + (size_t)(r0 + d) * A.row_bytes + off));
          }
      }
#pragma unroll
      for (int d = 0; d < Depth; ++d) acc += v[d];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) A.sink[blockIdx.x] = acc.x;
}

// mode 1: rows by LDS-DMA, the whole window in flight
__global__ __launch_bounds__(64) void rows_to_lds(const Args A) {
  const int lane = (int)threadIdx.x;
  const int tx = (int)blockIdx.x % A.tiles_x, ty = (int)blockIdx.x / A.tiles_x;
  const int chunks = (A.win_bytes + 1023) / 1024;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)s_dyn;
  float acc = 0.0f;
  for (int w = 0; w < A.windows; ++w) {
    const int y0 = (ty * A.windows + w) * A.advance;
    if (y0 + A.win_rows > A.rows) break;
    const char *base = A.src + (size_t)y0 * A.row_bytes + (size_t)tx * A.win_bytes + A.align_off;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int r = 0; r < A.win_rows; ++r)
      for (int c = 0; c < chunks; ++c) {
        const unsigned off = (unsigned)(c * 1024 + lane * 16);
        if ((int)off < A.win_bytes) {
          const char *row = base + (size_t)r * A.row_bytes;
          const unsigned lds = lds0 + (unsigned)(r * ((A.win_bytes + 15) & ~15) + c * 1024);
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(off), "s"(row) : "memory");
        }
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += reinterpret_cast<const float *>(s_dyn)[lane * 4 + (w & 3)];
  }
  if (acc == 12345.678f) A.sink[blockIdx.x] = acc;
}

// mode 2: lane = pixel of a 16 x 4 pass; per pixel 4 tap rows x (4 texels of texel_bytes) as dwordx4 loads
template <int TexelBytes> __global__ __launch_bounds__(64) void pixel_gathers(const Args A, int px_stride_texels, int row_stride) {
  const int lane = (int)threadIdx.x, pcol = lane & 15, prow = lane >> 4;
  const int tx = (int)blockIdx.x % A.tiles_x, ty = (int)blockIdx.x / A.tiles_x;
  constexpr int kLoads = (4 * TexelBytes + 15) / 16; // dwordx4 per tap row: 4 (RGBA) or 5 (RGBAZ)
  v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int w = 0; w < A.windows; ++w) {
    const int y0 = (ty * A.windows + w) * A.advance + prow * row_stride;
    if (y0 + 4 > A.rows) break;
    const char *p = A.src + (size_t)y0 * A.row_bytes + (size_t)tx * A.win_bytes + (size_t)(pcol * px_stride_texels) * TexelBytes + A.align_off;
    v4f t[4][kLoads];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < kLoads; ++i) t[j][i] = *reinterpret_cast<const v4f *>(p + (size_t)j * A.row_bytes + 16 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < kLoads; ++i) acc += t[j][i];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) A.sink[blockIdx.x] = acc.x;
}

int main(int argc, char **argv) {
  int W = 4096, H = 4096, images = 4, reps = 5;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--size")) W = H = atoi(argv[i + 1]);
    if (!strcmp(argv[i], "--images")) images = atoi(argv[i + 1]);
    if (!strcmp(argv[i], "--reps")) reps = atoi(argv[i + 1]);
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::printf("# %s, %d CUs; image %d x %d; %d images rotating\n", prop.name, cus, W, H, images);
  const size_t max_row = (size_t)W * 20;
  std::vector<char *> src((size_t)images);
  for (auto &p : src) {
    CK(hipMalloc(&p, max_row * H + 4096));
    CK(hipMemset(p, 1, max_row * H + 4096));
  }
  float *sink;
  CK(hipMalloc(&sink, sizeof(float) * (1 << 22)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  auto run = [&](const char *label, int mode, int texel, int win_bytes, int win_rows, int advance, int windows, int waves_per_cu, int depth,
                 int align_off, int px_stride = 4, int row_stride = 2) -> int {
    Args A{};
    A.row_bytes = (unsigned)(W * texel);
    A.rows = H;
    A.win_rows = win_rows;
    A.win_bytes = win_bytes;
    A.advance = advance;
    A.windows = windows;
    A.tiles_x = (int)((size_t)A.row_bytes / (size_t)win_bytes);
    if ((size_t)A.tiles_x * win_bytes + align_off + 64 > A.row_bytes) A.tiles_x -= 1;
    A.align_off = align_off;
    A.sink = sink;
    const int tiles_y = H / (advance * windows);
    const unsigned grid = (unsigned)(A.tiles_x * tiles_y);
    // LDS per workgroup limits the wavefronts per CU: 160 KiB / waves (mode 1 also needs room for its window)
    size_t lds = (size_t)(160 * 1024) / (size_t)waves_per_cu;
    lds = lds / 1024 * 1024;
    if (lds > 64 * 1024) lds = 64 * 1024;
    const size_t need = mode == 1 ? (size_t)win_rows * (size_t)((win_bytes + 15) & ~15) + 1024 : 0;
    if (need > lds) {
      std::printf("%-34s window does not fit %zu bytes of LDS at %d waves/CU\n", label, lds, waves_per_cu);
      return 0;
    }
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
      A.src = src[(size_t)r % src.size()];
      CK(hipEventRecord(e0));
      if (mode == 0) {
        if (depth == 1) hipLaunchKernelGGL(rows_to_regs<1>, dim3(grid), dim3(64), lds, 0, A);
        else if (depth == 2) hipLaunchKernelGGL(rows_to_regs<2>, dim3(grid), dim3(64), lds, 0, A);
        else if (depth == 4) hipLaunchKernelGGL(rows_to_regs<4>, dim3(grid), dim3(64), lds, 0, A);
        else if (depth == 8) hipLaunchKernelGGL(rows_to_regs<8>, dim3(grid), dim3(64), lds, 0, A);
        else hipLaunchKernelGGL(rows_to_regs<16>, dim3(grid), dim3(64), lds, 0, A);
      } else if (mode == 1) {
        hipLaunchKernelGGL(rows_to_lds, dim3(grid), dim3(64), lds, 0, A);
      } else if (texel == 20) {
        hipLaunchKernelGGL(pixel_gathers<20>, dim3(grid), dim3(64), lds, 0, A, px_stride, row_stride);
      } else {
        hipLaunchKernelGGL(pixel_gathers<16>, dim3(grid), dim3(64), lds, 0, A, px_stride, row_stride);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms < best) best = ms;
    }
    const double windows_total = (double)grid * windows;
    const double bytes = mode == 2 ? windows_total * 64.0 * 4.0 * 4.0 * texel : windows_total * win_rows * (double)win_bytes;
    // 128-byte lines requested per window, offered concurrency per CU (lines a wavefront has requested before it waits)
    const double lines_row = (win_bytes + 127) / 128 + (align_off % 128 ? 1 : 0);
    const double inflight = mode == 0 ? depth * lines_row : mode == 1 ? win_rows * lines_row : 4.0 * 4.0 * lines_row; // per wavefront
    std::printf("%-34s T=%2d L=%4d R=%2d adv=%2d x%d  %2d waves/CU  %7.1f us  %7.1f GB/s useful  (%.0f lines offered per CU)\n", label, texel,
                win_bytes, win_rows, advance, windows, waves_per_cu, best * 1e3, bytes / (best * 1e-3) / 1e9, inflight * waves_per_cu);
    return 0;
  };

  // 1. the pass-window shape of configs[3] (RGBAZ: 67 texels x 20 B = 1340 B -> 1280 B here so that windows tile the row), 11 rows, next pass 8 rows down
  for (int waves : {4, 8, 12, 16, 24, 32})
    for (int depth : {1, 2, 4, 8, 16}) run("regs  pass window RGBAZ", 0, 20, 1280, 11, 8, 4, waves, depth, 0);
  for (int waves : {4, 8, 12, 16}) run("lds   pass window RGBAZ", 1, 20, 1280, 11, 8, 4, waves, 0, 0);
  // 2. the same bytes as the direct path requests them (lane = pixel, 80-byte pixel stride, rows of a pass 2 apart)
  for (int waves : {4, 8, 12, 16, 24, 32}) run("gather direct path RGBAZ", 2, 20, 1280, 4, 8, 4, waves, 0, 0);
  for (int waves : {8, 16, 32}) run("gather direct path RGBA", 2, 16, 1024, 4, 8, 4, waves, 0, 0);
  // 3. row length and alignment (RGBA rows, 16 waves per CU, 8 rows in flight)
  for (int L : {64, 128, 256, 512, 1024})
    for (int off : {0, 64}) run(off ? "regs  rows +64 B" : "regs  rows aligned", 0, 16, L, 10, 8, 4, 16, 8, off);
  // 4. no halo (every row fetched once by one wavefront): the plain streaming rate of this access shape
  for (int waves : {8, 16, 32}) run("regs  no halo", 0, 20, 1280, 8, 8, 4, waves, 8, 0);
  for (int waves : {8, 16}) run("lds   no halo", 1, 20, 1280, 8, 8, 4, waves, 0, 0);
  // 5. whole-block windows (35 rows: the halo between passes disappears; 2 waves per SIMD have the LDS for it)
  for (int waves : {4, 8}) run("lds   block window RGBAZ (35 rows)", 1, 20, 1280, 35, 32, 1, waves, 0, 0);
  for (int waves : {8, 16}) run("regs  block window RGBAZ (35 rows)", 0, 20, 1280, 35, 32, 1, waves, 8, 0);
  return 0;
}
