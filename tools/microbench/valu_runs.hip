// valu_runs.hip — does a wavefront's VALU stream pay for SWITCHING between the 2-cycle class (plain
// v_mul / v_add / v_sub / v_fma / v_mov with VGPR, inline or literal operands) and the 4-cycle class
// (v_pk_*_f32, SGPR operands, cvt, cndmask, ...), and if so, how long must a run of plain instructions
// be before they issue at 2 cycles again?  valu_operands.hip measured 8 packed + 8 plain at 4.15 cycles
// per instruction (as if the plain ones cost 4 as well); this one sweeps the run length.
// Every kernel: ITERS x [ A instructions of class X, then B plain VOP2 ], no dependences inside a run
// shorter than 8 instructions, 4 and 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_runs.hip -o valu_runs
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 512;
#define CLOB "vcc","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","s40","s41"
#define INIT "s_mov_b32 s40, 0x3f800001\n s_mov_b32 s41, 0x3f800001\n" \
  "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n" \
  "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n" \
  "v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0\n v_mov_b32 v36, 1.0\n v_mov_b32 v37, 1.0\n v_mov_b32 v38, 1.0\n v_mov_b32 v39, 1.0\n" \
  "v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v42, 1.0\n v_mov_b32 v43, 1.0\n v_mov_b32 v44, 1.0\n v_mov_b32 v45, 1.0\n v_mov_b32 v46, 1.0\n v_mov_b32 v47, 1.0\n" \
  "v_mov_b32 v48, 1.0\n v_mov_b32 v49, 1.0\n v_mov_b32 v50, 1.0\n v_mov_b32 v51, 1.0\n v_mov_b32 v52, 1.0\n v_mov_b32 v53, 1.0\n v_mov_b32 v54, 1.0\n v_mov_b32 v55, 1.0\n"

// 8 packed multiplies on v[16:31] (sources v[32:47])
#define PK8 "v_pk_mul_f32 v[16:17], v[16:17], v[32:33]\n v_pk_mul_f32 v[18:19], v[18:19], v[34:35]\n v_pk_mul_f32 v[20:21], v[20:21], v[36:37]\n v_pk_mul_f32 v[22:23], v[22:23], v[38:39]\n v_pk_mul_f32 v[24:25], v[24:25], v[40:41]\n v_pk_mul_f32 v[26:27], v[26:27], v[42:43]\n v_pk_mul_f32 v[28:29], v[28:29], v[44:45]\n v_pk_mul_f32 v[30:31], v[30:31], v[46:47]\n "
// 8 plain multiplies on v48..v55 (independent of the packed registers: no cross-class dependence)
#define PL8 "v_mul_f32 v48, v48, v33\n v_mul_f32 v49, v49, v34\n v_mul_f32 v50, v50, v35\n v_mul_f32 v51, v51, v36\n v_mul_f32 v52, v52, v37\n v_mul_f32 v53, v53, v38\n v_mul_f32 v54, v54, v39\n v_mul_f32 v55, v55, v40\n "
// 8 plain adds, same registers (a second flavour so that a run is not one opcode)
#define PA8 "v_add_f32 v48, v48, v41\n v_add_f32 v49, v49, v42\n v_add_f32 v50, v50, v43\n v_add_f32 v51, v51, v44\n v_add_f32 v52, v52, v45\n v_add_f32 v53, v53, v46\n v_add_f32 v54, v54, v47\n v_add_f32 v55, v55, v32\n "
// 8 multiplies by an SGPR (4-cycle class by operand kind)
#define SG8 "v_mul_f32 v16, s40, v16\n v_mul_f32 v17, s40, v17\n v_mul_f32 v18, s40, v18\n v_mul_f32 v19, s40, v19\n v_mul_f32 v20, s40, v20\n v_mul_f32 v21, s40, v21\n v_mul_f32 v22, s40, v22\n v_mul_f32 v23, s40, v23\n "
// 8 conversions (4-cycle class by opcode)
#define CV8 "v_cvt_i32_f32 v16, v33\n v_cvt_i32_f32 v17, v34\n v_cvt_i32_f32 v18, v35\n v_cvt_i32_f32 v19, v36\n v_cvt_i32_f32 v20, v37\n v_cvt_i32_f32 v21, v38\n v_cvt_i32_f32 v22, v39\n v_cvt_i32_f32 v23, v40\n "
// ONE packed instruction / ONE plain
#define PK1 "v_pk_mul_f32 v[16:17], v[16:17], v[32:33]\n "
#define PK1B "v_pk_mul_f32 v[18:19], v[18:19], v[34:35]\n "
#define PL1 "v_mul_f32 v48, v48, v33\n "
#define PL1B "v_mul_f32 v49, v49, v34\n "
#define PL1C "v_mul_f32 v50, v50, v35\n "

#define KERNEL(ID, BODY)                                                                   \
  __global__ __launch_bounds__(256) void k##ID(float *out) {                               \
    asm volatile(INIT ::: CLOB);                                                           \
    for (int i = 0; i < ITERS; ++i) asm volatile(BODY ::: CLOB);                           \
    float r;                                                                               \
    asm volatile("v_add_f32 %0, v16, v48\n v_add_f32 %0, %0, v18\n v_add_f32 %0, %0, v49" : "=v"(r) :: CLOB); \
    if (r == 12345.678f) out[0] = r;                                                       \
  }
KERNEL(0, PL8 PA8 PL8 PA8)                             // 32 plain
KERNEL(1, PK8 PK8 PK8 PK8)                             // 32 packed
KERNEL(2, PK8 PL8)                                     // 8 + 8
KERNEL(3, PK8 PL8 PA8)                                 // 8 + 16
KERNEL(4, PK8 PL8 PA8 PL8 PA8)                         // 8 + 32
KERNEL(5, PK8 PL8 PA8 PL8 PA8 PL8 PA8 PL8 PA8)         // 8 + 64
KERNEL(6, PK8 PK8 PK8 PK8 PL8 PA8 PL8 PA8 PL8 PA8 PL8 PA8) // 32 + 64
KERNEL(7, SG8 PL8 PA8 PL8 PA8)                         // 8 sgpr-operand + 32 plain
KERNEL(8, CV8 PL8 PA8 PL8 PA8)                         // 8 cvt + 32 plain
KERNEL(9, PK1 PL1 PL1B PK1B PL1C PL1 PK1 PL1B PL1C PK1B PL1 PL1B) // packed, plain, plain, ... (4 pk + 8 plain)
KERNEL(10, PK1 PL8 PA8 PL8 PA8)                        // 1 + 32
KERNEL(11, PK1 PK1B PL8 PA8 PL8 PA8 PL8 PA8 PL8 PA8)   // 2 + 64

typedef void (*kfn)(float *);
struct Test { const char *name; kfn f; int n4, n2; }; // instructions of the 4-cycle class and of the 2-cycle class per iteration
int run(const Test &t, int waves_per_simd) {
  float *out; CHECK(hipMalloc(&out, 4));
  const int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, out);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 10; ++r) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, out);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double per_simd = (double)waves_per_simd * ITERS; // iterations per SIMD
  const double cyc_iter = best * 1e-3 * 2.4e9 / per_simd;
  const double ideal = 4.0 * t.n4 + 2.0 * t.n2;
  printf("%-44s waves/SIMD %d: %7.3f ms  %6.1f cycles per iteration (%d x 4 + %d x 2 = %.0f if classes do not interact; %.2f per instruction)\n",
         t.name, waves_per_simd, best, cyc_iter, t.n4, t.n2, ideal, cyc_iter / (t.n4 + t.n2));
  CHECK(hipFree(out));
  return 0;
}
int main() {
  const Test tests[] = {{"32 plain", k0, 0, 32}, {"32 packed", k1, 32, 0}, {"8 packed + 8 plain", k2, 8, 8}, {"8 packed + 16 plain", k3, 8, 16},
                        {"8 packed + 32 plain", k4, 8, 32}, {"8 packed + 64 plain", k5, 8, 64}, {"32 packed + 64 plain", k6, 32, 64},
                        {"8 sgpr-operand + 32 plain", k7, 8, 32}, {"8 cvt + 32 plain", k8, 8, 32}, {"(pk, plain, plain) x 4", k9, 4, 8},
                        {"1 packed + 32 plain", k10, 1, 32}, {"2 packed + 64 plain", k11, 2, 64}};
  for (int w : {4, 8})
    for (const Test &t : tests)
      if (run(t, w)) return 1;
  return 0;
}
