// store_stride.hip — what a 20-byte pixel stride costs on the store side (RGBAZ output).
// A wavefront writes 64 consecutive pixels of an output row = 1280 contiguous bytes, either
//   A: as the kernels do today: one dwordx4 + one dword per lane at a 20-byte lane stride
//      (two instructions that each touch every 64-byte segment of the run partially);
//   B: as 80 aligned 16-byte chunks (64 lanes, then 16 lanes): every segment written whole —
//      the values exchanged through LDS first (lane i writes its 5 floats at 20 i, reads 16 i);
//   C: B without the LDS exchange (synthetic values, the store pattern alone);
//   D: RGBA reference: one dwordx4 per lane at a 16-byte stride, the same number of pixels.
// All stores non-temporal, buffers rotate over more memory than the 256 MB Infinity Cache.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/store_stride.hip -o tools/microbench/store_stride
#include <hip/hip_runtime.h>

#include <cstdio>
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
typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kRowsPerWave = 16; // rows of 64 pixels one wavefront writes (like a tile of the kernels)

template <int Mode> __global__ __launch_bounds__(64) void store_kernel(float *dst, int width, int height, float seed) {
  __shared__ float s_x[64 * 5];
  const int lane = (int)threadIdx.x;
  const int tiles_x = width / 64;
  const int tx = (int)blockIdx.x % tiles_x, ty = (int)blockIdx.x / tiles_x;
  for (int r = 0; r < kRowsPerWave; ++r) {
    const int y = ty * kRowsPerWave + r;
    if (y >= height) return;
    const size_t px = (size_t)y * width + (size_t)tx * 64;
    const float c0 = seed + (float)lane, c1 = c0 * 0.5f, c2 = c0 + 1.0f, c3 = c0 - (float)r, c4 = c1 + (float)r;
    if constexpr (Mode == 0) { // A
      float *d = dst + (px + lane) * 5;
      __builtin_nontemporal_store(v4f_a4{c0, c1, c2, c3}, reinterpret_cast<v4f_a4 *>(d));
      __builtin_nontemporal_store(c4, d + 4);
    } else if constexpr (Mode == 1) { // B
      float *x = s_x + lane * 5;
      x[0] = c0; x[1] = c1; x[2] = c2; x[3] = c3; x[4] = c4;
      __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): single wavefront, no barrier needed
      const v4f q0 = *reinterpret_cast<const v4f *>(s_x + lane * 4);
      v4f q1 = v4f{0.0f, 0.0f, 0.0f, 0.0f};
      if (lane < 16) q1 = *reinterpret_cast<const v4f *>(s_x + 256 + lane * 4);
      float *d = dst + px * 5;
      __builtin_nontemporal_store(q0, reinterpret_cast<v4f *>(d + lane * 4));
      if (lane < 16) __builtin_nontemporal_store(q1, reinterpret_cast<v4f *>(d + 256 + lane * 4));
      __builtin_amdgcn_s_waitcnt(0xc07f);
    } else if constexpr (Mode == 2) { // C
      float *d = dst + px * 5;
      __builtin_nontemporal_store(v4f{c0, c1, c2, c3}, reinterpret_cast<v4f *>(d + lane * 4));
      if (lane < 16) __builtin_nontemporal_store(v4f{c4, c1, c2, c3}, reinterpret_cast<v4f *>(d + 256 + lane * 4));
    } else { // D
      float *d = dst + (px + lane) * 4;
      __builtin_nontemporal_store(v4f{c0, c1, c2, c3}, reinterpret_cast<v4f *>(d));
    }
  }
}

int main() {
  const int w = 4096, h = 4096, n_buf = 4, reps = 40;
  std::vector<float *> bufs(n_buf);
  for (auto &b : bufs) CK(hipMalloc((void **)&b, (size_t)w * h * 5 * sizeof(float)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((w / 64) * ((h + kRowsPerWave - 1) / kRowsPerWave));
  const char *names[4] = {"A dwordx4 + dword at a 20-byte stride", "B 16-byte chunks, through LDS", "C 16-byte chunks, pattern only",
                          "D RGBA dwordx4 at a 16-byte stride"};
  for (int mode = 0; mode < 4; ++mode) {
    auto launch = [&](int i) {
      float *d = bufs[i % n_buf];
      if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(64), 0, 0, d, w, h, (float)i);
      if (mode == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(64), 0, 0, d, w, h, (float)i);
      if (mode == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(grid), dim3(64), 0, 0, d, w, h, (float)i);
      if (mode == 3) hipLaunchKernelGGL(store_kernel<3>, dim3(grid), dim3(64), 0, 0, d, w, h, (float)i);
    };
    for (int i = 0; i < 8; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = (double)w * h * (mode == 3 ? 16 : 20);
    std::printf("%-40s %7.1f us per 4096^2 frame  %6.2f TB/s\n", names[mode], us, bytes / us / 1e6);
  }
  for (auto &b : bufs) CK(hipFree(b));
  return 0;
}
