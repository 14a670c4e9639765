// cubic_rate.hip — cost of the reference's bicubic arithmetic (20 Catmull-Rom
// evaluations per RGBA pixel, un-fused) on register data, packed (v_pk_*) versus
// scalar, as a function of resident wavefronts per SIMD.
// CAVEAT (found later in the round): only 4 of the 32 tap pairs change per iteration, so the
// compiler hoists the weight-independent 11 of 17 operations of the other cubics out of the
// loop; the cycles printed here are for ~122 of the 170 packed instructions.  Instruction
// costs proper: valu_operands.hip.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off cubic_rate.hip -o cubic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T> __device__ __forceinline__ T cr(T a, T b, T c, T d, float t, float ht) {
  const T inner = ((3.0f * (b - c)) + d) - a;
  const T mid = ((((2.0f * a) - (5.0f * b)) + (4.0f * c)) - d) + t * inner;
  const T outer = (c - a) + t * mid;
  return b + ht * outer;
}

constexpr int ITERS = 512;

template <bool Packed> __global__ __launch_bounds__(256) void k(float *out, float seed, int lds_pad) {
  extern __shared__ float pad[];
  if (lds_pad < 0) pad[threadIdx.x] = seed; // never
  float fx = seed * 0.25f + threadIdx.x * 1e-3f, fy = seed * 0.5f;
  float hfx = 0.5f * fx, hfy = 0.5f * fy;
  if constexpr (Packed) {
    f2 p[16][2];
    for (int i = 0; i < 16; ++i) { p[i][0] = f2{seed + i, seed - i}; p[i][1] = f2{seed * i, seed + 2 * i}; }
    f2 r0 = {0, 0}, r1 = {0, 0};
    for (int it = 0; it < ITERS; ++it) {
      f2 c[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        c[i][0] = cr<f2>(p[i][0], p[4 + i][0], p[8 + i][0], p[12 + i][0], fy, hfy);
        c[i][1] = cr<f2>(p[i][1], p[4 + i][1], p[8 + i][1], p[12 + i][1], fy, hfy);
      }
      f2 a = cr<f2>(c[0][0], c[1][0], c[2][0], c[3][0], fx, hfx);
      f2 b = cr<f2>(c[0][1], c[1][1], c[2][1], c[3][1], fx, hfx);
      r0 += a; r1 += b;
      // perturb the taps so nothing is loop-invariant
      p[0][0] = a; p[5][1] = b; p[10][0] = p[10][0] + a; p[15][1] = p[15][1] + b;
      asm volatile("" : "+v"(fx), "+v"(fy));
    }
    if (r0.x + r0.y + r1.x + r1.y == 12345.678f) out[0] = r0.x;
  } else {
    float p[16][4];
    for (int i = 0; i < 16; ++i) { p[i][0] = seed + i; p[i][1] = seed - i; p[i][2] = seed * i; p[i][3] = seed + 2 * i; }
    float r[4] = {0, 0, 0, 0};
    for (int it = 0; it < ITERS; ++it) {
      float c[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) c[i][ch] = cr<float>(p[i][ch], p[4 + i][ch], p[8 + i][ch], p[12 + i][ch], fy, hfy);
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        float a = cr<float>(c[0][ch], c[1][ch], c[2][ch], c[3][ch], fx, hfx);
        r[ch] += a;
        p[ch * 5][ch] = a;
      }
      asm volatile("" : "+v"(fx), "+v"(fy));
    }
    if (r[0] + r[1] + r[2] + r[3] == 12345.678f) out[0] = r[0];
  }
}

template <bool Packed> int run(const char *name, int waves_per_simd) {
  float *out; CHECK(hipMalloc(&out, 4));
  // occupancy control through dynamic LDS: 160 KiB / CU, 256-thread blocks = 1 wave per SIMD each
  const int lds = (160 * 1024) / waves_per_simd - 512;
  CHECK(hipFuncSetAttribute((const void *)k<Packed>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  const int blocks = 256 * waves_per_simd * 4;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<Packed>, dim3(blocks), dim3(256), lds, 0, out, 1.0f, 0);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<Packed>, dim3(blocks), dim3(256), lds, 0, out, 1.0f, 0);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double bicubics_per_simd = (double)blocks * 4 * ITERS / 1024.0;
  printf("%-8s waves/SIMD %d: %8.3f ms  %7.1f cycles per wave-bicubic per SIMD (@2.4GHz)  = %6.1f us per 4K frame\n", name,
         waves_per_simd, best, best * 1e-3 * 2.4e9 / bicubics_per_simd, 262144.0 / 1024.0 * (best * 1e-3 / bicubics_per_simd) * 1e6);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  for (int w : {1, 2, 3, 4, 6, 8}) { run<true>("packed", w); run<false>("scalar", w); }
  return 0;
}
