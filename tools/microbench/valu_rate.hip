// valu_rate.hip — measures the per-instruction VALU issue rates the kernel
// design depends on (gfx950): plain vs packed f32 mul/add, f64 fma, v_rcp_f32,
// the compiler's IEEE f32 divide and sqrt sequences.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

template <int MODE> __global__ __launch_bounds__(256) void k(float *out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float kf = 1.0000001f;
  if constexpr (MODE == 0) { // v_mul_f32 x8
    for (int i = 0; i < ITERS; ++i) {
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(kf));
    }
  } else if constexpr (MODE == 1) { // v_pk_mul_f32 x4 (8 floats)
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    f2 kk = {kf, kf};
    for (int i = 0; i < ITERS; ++i) {
      asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(kk));
    }
    a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
  } else if constexpr (MODE == 2) { // v_pk_add_f32 x8 instr
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    f2 kk = {kf, kf};
    for (int i = 0; i < ITERS; ++i) {
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(kk));
    }
    a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
  } else if constexpr (MODE == 3) { // v_fma_f64 x8
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, kd = 1.0000001, c = 1e-9;
    for (int i = 0; i < ITERS; ++i) {
      asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                   "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(kd), "v"(c));
    }
    a0 = d0; a1 = d1; a2 = d2; a3 = d3; a4 = d4; a5 = d5; a6 = d6; a7 = d7;
  } else if constexpr (MODE == 4) { // v_rcp_f32 x8
    for (int i = 0; i < ITERS; ++i) {
      asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                   "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
  } else if constexpr (MODE == 5) { // IEEE divide x8 (compiler sequence)
    float b = seed + 1.5f;
    for (int i = 0; i < ITERS; ++i) {
      a0 = a0 / b; a1 = a1 / b; a2 = a2 / b; a3 = a3 / b; a4 = a4 / b; a5 = a5 / b; a6 = a6 / b; a7 = a7 / b;
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b));
    }
  } else if constexpr (MODE == 6) { // IEEE sqrt x8
    for (int i = 0; i < ITERS; ++i) {
      a0 = __builtin_sqrtf(a0); a1 = __builtin_sqrtf(a1); a2 = __builtin_sqrtf(a2); a3 = __builtin_sqrtf(a3);
      a4 = __builtin_sqrtf(a4); a5 = __builtin_sqrtf(a5); a6 = __builtin_sqrtf(a6); a7 = __builtin_sqrtf(a7);
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
  } else if constexpr (MODE == 7) { // v_fma_f32 x8
    for (int i = 0; i < ITERS; ++i) {
      asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(kf));
    }
  } else if constexpr (MODE == 8) { // v_pk_fma_f32 x8 instr
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    f2 kk = {kf, kf};
    for (int i = 0; i < ITERS; ++i) {
      asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                   "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(kk));
    }
    a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s == 12345.678f) out[0] = s;
}

template <int MODE> int run(const char *name, int instr_per_iter, int blocks) {
  float *out;
  CHECK(hipMalloc(&out, 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double wave_instr = (double)blocks * 4 * ITERS * instr_per_iter;
  double t = best * 1e-3;
  // cycles per wave-instruction per SIMD at 2.4 GHz, 1024 SIMDs
  double cyc = t * 2.4e9 * 1024.0 / wave_instr;
  printf("%-28s blocks=%5d  %8.3f ms  %7.2f T lane-instr/s  %5.2f cyc/wave-instr/SIMD (@2.4GHz)\n", name, blocks, best,
         wave_instr * 64 / t / 1e12, cyc);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  for (int blocks : {256, 2048}) {
    run<0>("v_mul_f32", 8, blocks);
    run<7>("v_fma_f32", 8, blocks);
    run<1>("v_pk_mul_f32", 8, blocks);
    run<2>("v_pk_add_f32", 8, blocks);
    run<8>("v_pk_fma_f32", 8, blocks);
    run<3>("v_fma_f64", 8, blocks);
    run<4>("v_rcp_f32", 8, blocks);
    run<5>("ieee div f32 (per div)", 8, blocks);
    run<6>("ieee sqrt f32 (per sqrt)", 8, blocks);
  }
  return 0;
}
