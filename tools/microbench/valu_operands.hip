// valu_operands.hip — issue cost of f32 VALU instructions on gfx950 as a function of
// encoding (VOP2 / VOP3 / VOP3P), operand kinds (VGPR / SGPR / inline / literal) and
// VGPR bank placement.  Fixed registers in inline asm so the compiler cannot move things.
// Build: hipcc --offload-arch=gfx950 -O3 valu_operands.hip -o valu_operands
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 1024;

#define CLOB "vcc","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","s40","s41","s42","s43"

#define INIT "s_mov_b32 s40, 0x3f800001\n s_mov_b32 s41, 0x3f800001\n s_mov_b32 s42, 0x3f7fffff\n s_mov_b32 s43, 0x3f7fffff\n" \
  "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n" \
  "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n" \
  "v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0\n v_mov_b32 v36, 1.0\n v_mov_b32 v37, 1.0\n v_mov_b32 v38, 1.0\n v_mov_b32 v39, 1.0\n" \
  "v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v42, 1.0\n v_mov_b32 v43, 1.0\n v_mov_b32 v44, 1.0\n v_mov_b32 v45, 1.0\n v_mov_b32 v46, 1.0\n v_mov_b32 v47, 1.0\n"

// every body is 8 instructions; the loop runs it twice per iteration
#define TEST(ID, BODY)                                                                     \
  __global__ __launch_bounds__(256) void t##ID(float *out) {                               \
    asm volatile(INIT ::: CLOB);                                                           \
    for (int i = 0; i < ITERS; ++i) asm volatile(BODY BODY ::: CLOB);                      \
    float r;                                                                               \
    asm volatile("v_add_f32 %0, v16, v17\n v_add_f32 %0, %0, v18\n v_add_f32 %0, %0, v20\n v_add_f32 %0, %0, v24" : "=v"(r) :: CLOB); \
    if (r == 12345.678f) out[0] = r;                                                       \
  }

// 1: VOP2 mul, two VGPRs, dst chains of length 8 (v16..v23), second source v32.. (different banks: 16%4=0 vs 33%4=1)
TEST(1, "v_mul_f32 v16, v16, v33\n v_mul_f32 v17, v17, v34\n v_mul_f32 v18, v18, v35\n v_mul_f32 v19, v19, v36\n v_mul_f32 v20, v20, v37\n v_mul_f32 v21, v21, v38\n v_mul_f32 v22, v22, v39\n v_mul_f32 v23, v23, v40\n")
// 2: same, sources in the SAME bank (16 and 32, 17 and 33 ...)
TEST(2, "v_mul_f32 v16, v16, v32\n v_mul_f32 v17, v17, v33\n v_mul_f32 v18, v18, v34\n v_mul_f32 v19, v19, v35\n v_mul_f32 v20, v20, v36\n v_mul_f32 v21, v21, v37\n v_mul_f32 v22, v22, v38\n v_mul_f32 v23, v23, v39\n")
// 3: VOP2 mul by SGPR
TEST(3, "v_mul_f32 v16, s40, v16\n v_mul_f32 v17, s40, v17\n v_mul_f32 v18, s40, v18\n v_mul_f32 v19, s40, v19\n v_mul_f32 v20, s40, v20\n v_mul_f32 v21, s40, v21\n v_mul_f32 v22, s40, v22\n v_mul_f32 v23, s40, v23\n")
// 4: VOP2 mul by inline constant
TEST(4, "v_mul_f32 v16, 1.0, v16\n v_mul_f32 v17, 1.0, v17\n v_mul_f32 v18, 1.0, v18\n v_mul_f32 v19, 1.0, v19\n v_mul_f32 v20, 1.0, v20\n v_mul_f32 v21, 1.0, v21\n v_mul_f32 v22, 1.0, v22\n v_mul_f32 v23, 1.0, v23\n")
// 5: VOP2 mul by literal (8-byte instruction)
TEST(5, "v_mul_f32 v16, 0x3f800001, v16\n v_mul_f32 v17, 0x3f800001, v17\n v_mul_f32 v18, 0x3f800001, v18\n v_mul_f32 v19, 0x3f800001, v19\n v_mul_f32 v20, 0x3f800001, v20\n v_mul_f32 v21, 0x3f800001, v21\n v_mul_f32 v22, 0x3f800001, v22\n v_mul_f32 v23, 0x3f800001, v23\n")
// 6: VOP3-encoded mul, two VGPRs (8-byte instruction, same operands as test 1)
TEST(6, "v_mul_f32_e64 v16, v16, v33\n v_mul_f32_e64 v17, v17, v34\n v_mul_f32_e64 v18, v18, v35\n v_mul_f32_e64 v19, v19, v36\n v_mul_f32_e64 v20, v20, v37\n v_mul_f32_e64 v21, v21, v38\n v_mul_f32_e64 v22, v22, v39\n v_mul_f32_e64 v23, v23, v40\n")
// 7: v_fma_f32 three VGPRs
TEST(7, "v_fma_f32 v16, v16, v33, v42\n v_fma_f32 v17, v17, v34, v43\n v_fma_f32 v18, v18, v35, v44\n v_fma_f32 v19, v19, v36, v45\n v_fma_f32 v20, v20, v37, v46\n v_fma_f32 v21, v21, v38, v47\n v_fma_f32 v22, v22, v39, v44\n v_fma_f32 v23, v23, v40, v45\n")
// 8: v_fma_f32 two VGPRs + SGPR
TEST(8, "v_fma_f32 v16, v16, s40, v42\n v_fma_f32 v17, v17, s40, v43\n v_fma_f32 v18, v18, s40, v44\n v_fma_f32 v19, v19, s40, v45\n v_fma_f32 v20, v20, s40, v46\n v_fma_f32 v21, v21, s40, v47\n v_fma_f32 v22, v22, s40, v44\n v_fma_f32 v23, v23, s40, v45\n")
// 9: v_pk_mul_f32 two VGPR pairs
TEST(9, "v_pk_mul_f32 v[16:17], v[16:17], v[32:33]\n v_pk_mul_f32 v[18:19], v[18:19], v[34:35]\n v_pk_mul_f32 v[20:21], v[20:21], v[36:37]\n v_pk_mul_f32 v[22:23], v[22:23], v[38:39]\n v_pk_mul_f32 v[24:25], v[24:25], v[40:41]\n v_pk_mul_f32 v[26:27], v[26:27], v[42:43]\n v_pk_mul_f32 v[28:29], v[28:29], v[44:45]\n v_pk_mul_f32 v[30:31], v[30:31], v[46:47]\n")
// 10: v_pk_mul_f32 VGPR pair x SGPR pair
TEST(10, "v_pk_mul_f32 v[16:17], v[16:17], s[40:41]\n v_pk_mul_f32 v[18:19], v[18:19], s[40:41]\n v_pk_mul_f32 v[20:21], v[20:21], s[40:41]\n v_pk_mul_f32 v[22:23], v[22:23], s[40:41]\n v_pk_mul_f32 v[24:25], v[24:25], s[40:41]\n v_pk_mul_f32 v[26:27], v[26:27], s[40:41]\n v_pk_mul_f32 v[28:29], v[28:29], s[40:41]\n v_pk_mul_f32 v[30:31], v[30:31], s[40:41]\n")
// 11: v_pk_add_f32 two VGPR pairs, sources two registers apart in bank order
TEST(11, "v_pk_add_f32 v[16:17], v[16:17], v[34:35]\n v_pk_add_f32 v[18:19], v[18:19], v[36:37]\n v_pk_add_f32 v[20:21], v[20:21], v[38:39]\n v_pk_add_f32 v[22:23], v[22:23], v[40:41]\n v_pk_add_f32 v[24:25], v[24:25], v[42:43]\n v_pk_add_f32 v[26:27], v[26:27], v[44:45]\n v_pk_add_f32 v[28:29], v[28:29], v[46:47]\n v_pk_add_f32 v[30:31], v[30:31], v[34:35]\n")
// 12: v_pk_mul_f32 with the broadcast scalar weight in one VGPR (op_sel_hi = 0 on src1: both halves read v32)
TEST(12, "v_pk_mul_f32 v[16:17], v[16:17], v[32:33] op_sel_hi:[1,0]\n v_pk_mul_f32 v[18:19], v[18:19], v[32:33] op_sel_hi:[1,0]\n v_pk_mul_f32 v[20:21], v[20:21], v[32:33] op_sel_hi:[1,0]\n v_pk_mul_f32 v[22:23], v[22:23], v[32:33] op_sel_hi:[1,0]\n v_pk_mul_f32 v[24:25], v[24:25], v[32:33] op_sel_hi:[1,0]\n v_pk_mul_f32 v[26:27], v[26:27], v[32:33] op_sel_hi:[1,0]\n v_pk_mul_f32 v[28:29], v[28:29], v[32:33] op_sel_hi:[1,0]\n v_pk_mul_f32 v[30:31], v[30:31], v[32:33] op_sel_hi:[1,0]\n")
// 13: VOP2 add (sub) two VGPRs, non-destructive (dst differs from sources)
TEST(13, "v_sub_f32 v16, v24, v33\n v_sub_f32 v17, v25, v34\n v_sub_f32 v18, v26, v35\n v_sub_f32 v19, v27, v36\n v_sub_f32 v20, v28, v37\n v_sub_f32 v21, v29, v38\n v_sub_f32 v22, v30, v39\n v_sub_f32 v23, v31, v40\n")
// 14: v_mov_b32 (VOP1)
TEST(14, "v_mov_b32 v16, v33\n v_mov_b32 v17, v34\n v_mov_b32 v18, v35\n v_mov_b32 v19, v36\n v_mov_b32 v20, v37\n v_mov_b32 v21, v38\n v_mov_b32 v22, v39\n v_mov_b32 v23, v40\n")
// 15: v_pk_fma_f32 three pairs
TEST(15, "v_pk_fma_f32 v[16:17], v[16:17], v[32:33], v[40:41]\n v_pk_fma_f32 v[18:19], v[18:19], v[34:35], v[42:43]\n v_pk_fma_f32 v[20:21], v[20:21], v[36:37], v[44:45]\n v_pk_fma_f32 v[22:23], v[22:23], v[38:39], v[46:47]\n v_pk_fma_f32 v[24:25], v[24:25], v[32:33], v[40:41]\n v_pk_fma_f32 v[26:27], v[26:27], v[34:35], v[42:43]\n v_pk_fma_f32 v[28:29], v[28:29], v[36:37], v[44:45]\n v_pk_fma_f32 v[30:31], v[30:31], v[38:39], v[46:47]\n")
// 16: v_cndmask_b32 VOP2 (vcc)
TEST(16, "v_cndmask_b32 v16, v16, v33, vcc\n v_cndmask_b32 v17, v17, v34, vcc\n v_cndmask_b32 v18, v18, v35, vcc\n v_cndmask_b32 v19, v19, v36, vcc\n v_cndmask_b32 v20, v20, v37, vcc\n v_cndmask_b32 v21, v21, v38, vcc\n v_cndmask_b32 v22, v22, v39, vcc\n v_cndmask_b32 v23, v23, v40, vcc\n")
// 17: mixed: one pk + two VOP2 interleaved (does the mix pipeline better than either?)
TEST(17, "v_pk_mul_f32 v[16:17], v[16:17], v[32:33]\n v_mul_f32 v24, v24, v41\n v_mul_f32 v25, v25, v42\n v_pk_mul_f32 v[18:19], v[18:19], v[34:35]\n v_mul_f32 v26, v26, v43\n v_mul_f32 v27, v27, v44\n v_pk_mul_f32 v[20:21], v[20:21], v[36:37]\n v_mul_f32 v28, v28, v45\n")


#define TEST2(ID, B1, B2)                                                                  \
  __global__ __launch_bounds__(256) void t##ID(float *out) {                               \
    asm volatile(INIT ::: CLOB);                                                           \
    for (int i = 0; i < ITERS; ++i) asm volatile(B1 B2 ::: CLOB);                   \
    float r;                                                                               \
    asm volatile("v_add_f32 %0, v16, v17" : "=v"(r) :: CLOB);                              \
    if (r == 12345.678f) out[0] = r;                                                       \
  }
// wave-level mix: even waves run packed multiplies, odd waves VOP2 multiplies
__global__ __launch_bounds__(256) void t49(float *out) {
  asm volatile(INIT ::: CLOB);
  if ((threadIdx.x >> 6) & 1) {
    for (int i = 0; i < ITERS; ++i) asm volatile("v_pk_mul_f32 v[16:17], v[16:17], v[32:33]\n v_pk_mul_f32 v[18:19], v[18:19], v[34:35]\n v_pk_mul_f32 v[20:21], v[20:21], v[36:37]\n v_pk_mul_f32 v[22:23], v[22:23], v[38:39]\n v_pk_mul_f32 v[24:25], v[24:25], v[40:41]\n v_pk_mul_f32 v[26:27], v[26:27], v[42:43]\n v_pk_mul_f32 v[28:29], v[28:29], v[44:45]\n v_pk_mul_f32 v[30:31], v[30:31], v[46:47]\n v_pk_mul_f32 v[16:17], v[16:17], v[32:33]\n v_pk_mul_f32 v[18:19], v[18:19], v[34:35]\n v_pk_mul_f32 v[20:21], v[20:21], v[36:37]\n v_pk_mul_f32 v[22:23], v[22:23], v[38:39]\n v_pk_mul_f32 v[24:25], v[24:25], v[40:41]\n v_pk_mul_f32 v[26:27], v[26:27], v[42:43]\n v_pk_mul_f32 v[28:29], v[28:29], v[44:45]\n v_pk_mul_f32 v[30:31], v[30:31], v[46:47]\n " ::: CLOB);
  } else {
    for (int i = 0; i < ITERS; ++i) asm volatile("v_mul_f32 v16, v16, v33\n v_mul_f32 v17, v17, v34\n v_mul_f32 v18, v18, v35\n v_mul_f32 v19, v19, v36\n v_mul_f32 v20, v20, v37\n v_mul_f32 v21, v21, v38\n v_mul_f32 v22, v22, v39\n v_mul_f32 v23, v23, v40\n v_mul_f32 v16, v16, v33\n v_mul_f32 v17, v17, v34\n v_mul_f32 v18, v18, v35\n v_mul_f32 v19, v19, v36\n v_mul_f32 v20, v20, v37\n v_mul_f32 v21, v21, v38\n v_mul_f32 v22, v22, v39\n v_mul_f32 v23, v23, v40\n " ::: CLOB);
  }
  float r;
  asm volatile("v_add_f32 %0, v16, v17" : "=v"(r) :: CLOB);
  if (r == 12345.678f) out[0] = r;
}
TEST(18, "v_mul_f32_e64 v16, v16, s40\n v_mul_f32_e64 v17, v17, s40\n v_mul_f32_e64 v18, v18, s40\n v_mul_f32_e64 v19, v19, s40\n v_mul_f32_e64 v20, v20, s40\n v_mul_f32_e64 v21, v21, s40\n v_mul_f32_e64 v22, v22, s40\n v_mul_f32_e64 v23, v23, s40\n ")
TEST(19, "v_add_f32 v16, s40, v16\n v_add_f32 v17, s40, v17\n v_add_f32 v18, s40, v18\n v_add_f32 v19, s40, v19\n v_add_f32 v20, s40, v20\n v_add_f32 v21, s40, v21\n v_add_f32 v22, s40, v22\n v_add_f32 v23, s40, v23\n ")
TEST(20, "v_cndmask_b32_e64 v16, v16, v33, s[42:43]\n v_cndmask_b32_e64 v17, v17, v34, s[42:43]\n v_cndmask_b32_e64 v18, v18, v35, s[42:43]\n v_cndmask_b32_e64 v19, v19, v36, s[42:43]\n v_cndmask_b32_e64 v20, v20, v37, s[42:43]\n v_cndmask_b32_e64 v21, v21, v38, s[42:43]\n v_cndmask_b32_e64 v22, v22, v39, s[42:43]\n v_cndmask_b32_e64 v23, v23, v40, s[42:43]\n ")
TEST(21, "v_cmp_gt_f32 vcc, v16, v33\n v_cmp_gt_f32 vcc, v17, v34\n v_cmp_gt_f32 vcc, v18, v35\n v_cmp_gt_f32 vcc, v19, v36\n v_cmp_gt_f32 vcc, v20, v37\n v_cmp_gt_f32 vcc, v21, v38\n v_cmp_gt_f32 vcc, v22, v39\n v_cmp_gt_f32 vcc, v23, v40\n ")
TEST(22, "v_cmp_gt_f32_e64 s[42:43], v16, v33\n v_cmp_gt_f32_e64 s[42:43], v17, v34\n v_cmp_gt_f32_e64 s[42:43], v18, v35\n v_cmp_gt_f32_e64 s[42:43], v19, v36\n v_cmp_gt_f32_e64 s[42:43], v20, v37\n v_cmp_gt_f32_e64 s[42:43], v21, v38\n v_cmp_gt_f32_e64 s[42:43], v22, v39\n v_cmp_gt_f32_e64 s[42:43], v23, v40\n ")
TEST(23, "v_cvt_i32_f32 v16, v33\n v_cvt_i32_f32 v17, v34\n v_cvt_i32_f32 v18, v35\n v_cvt_i32_f32 v19, v36\n v_cvt_i32_f32 v20, v37\n v_cvt_i32_f32 v21, v38\n v_cvt_i32_f32 v22, v39\n v_cvt_i32_f32 v23, v40\n ")
TEST(24, "v_trunc_f32 v16, v33\n v_trunc_f32 v17, v34\n v_trunc_f32 v18, v35\n v_trunc_f32 v19, v36\n v_trunc_f32 v20, v37\n v_trunc_f32 v21, v38\n v_trunc_f32 v22, v39\n v_trunc_f32 v23, v40\n ")
TEST(25, "v_rcp_f32 v16, v33\n v_rcp_f32 v17, v34\n v_rcp_f32 v18, v35\n v_rcp_f32 v19, v36\n v_rcp_f32 v20, v37\n v_rcp_f32 v21, v38\n v_rcp_f32 v22, v39\n v_rcp_f32 v23, v40\n ")
TEST(26, "v_sqrt_f32 v16, v33\n v_sqrt_f32 v17, v34\n v_sqrt_f32 v18, v35\n v_sqrt_f32 v19, v36\n v_sqrt_f32 v20, v37\n v_sqrt_f32 v21, v38\n v_sqrt_f32 v22, v39\n v_sqrt_f32 v23, v40\n ")
TEST(27, "v_div_scale_f32 v16, vcc, v16, v33, v16\n v_div_scale_f32 v17, vcc, v17, v34, v17\n v_div_scale_f32 v18, vcc, v18, v35, v18\n v_div_scale_f32 v19, vcc, v19, v36, v19\n v_div_scale_f32 v20, vcc, v20, v37, v20\n v_div_scale_f32 v21, vcc, v21, v38, v21\n v_div_scale_f32 v22, vcc, v22, v39, v22\n v_div_scale_f32 v23, vcc, v23, v40, v23\n ")
TEST(28, "v_div_fmas_f32 v16, v16, v33, v41\n v_div_fmas_f32 v17, v17, v34, v42\n v_div_fmas_f32 v18, v18, v35, v43\n v_div_fmas_f32 v19, v19, v36, v44\n v_div_fmas_f32 v20, v20, v37, v45\n v_div_fmas_f32 v21, v21, v38, v46\n v_div_fmas_f32 v22, v22, v39, v41\n v_div_fmas_f32 v23, v23, v40, v42\n ")
TEST(29, "v_div_fixup_f32 v16, v16, v33, v41\n v_div_fixup_f32 v17, v17, v34, v42\n v_div_fixup_f32 v18, v18, v35, v43\n v_div_fixup_f32 v19, v19, v36, v44\n v_div_fixup_f32 v20, v20, v37, v45\n v_div_fixup_f32 v21, v21, v38, v46\n v_div_fixup_f32 v22, v22, v39, v41\n v_div_fixup_f32 v23, v23, v40, v42\n ")
TEST(30, "v_add_u32 v16, v16, v33\n v_add_u32 v17, v17, v34\n v_add_u32 v18, v18, v35\n v_add_u32 v19, v19, v36\n v_add_u32 v20, v20, v37\n v_add_u32 v21, v21, v38\n v_add_u32 v22, v22, v39\n v_add_u32 v23, v23, v40\n ")
TEST(31, "v_lshlrev_b32 v16, 4, v33\n v_lshlrev_b32 v17, 4, v34\n v_lshlrev_b32 v18, 4, v35\n v_lshlrev_b32 v19, 4, v36\n v_lshlrev_b32 v20, 4, v37\n v_lshlrev_b32 v21, 4, v38\n v_lshlrev_b32 v22, 4, v39\n v_lshlrev_b32 v23, 4, v40\n ")
TEST(32, "v_and_b32 v16, 0x7fffffff, v33\n v_and_b32 v17, 0x7fffffff, v34\n v_and_b32 v18, 0x7fffffff, v35\n v_and_b32 v19, 0x7fffffff, v36\n v_and_b32 v20, 0x7fffffff, v37\n v_and_b32 v21, 0x7fffffff, v38\n v_and_b32 v22, 0x7fffffff, v39\n v_and_b32 v23, 0x7fffffff, v40\n ")
TEST(33, "v_mad_u32_u24 v16, v16, v33, v41\n v_mad_u32_u24 v17, v17, v34, v42\n v_mad_u32_u24 v18, v18, v35, v43\n v_mad_u32_u24 v19, v19, v36, v44\n v_mad_u32_u24 v20, v20, v37, v45\n v_mad_u32_u24 v21, v21, v38, v46\n v_mad_u32_u24 v22, v22, v39, v41\n v_mad_u32_u24 v23, v23, v40, v42\n ")
TEST(34, "v_mov_b32_dpp v16, v33 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v17, v34 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v18, v35 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v19, v36 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v20, v37 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v21, v38 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v22, v39 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v23, v40 row_shr:1 row_mask:0xf bank_mask:0xf\n ")
TEST(35, "v_max_i32 v16, v16, v33\n v_max_i32 v17, v17, v34\n v_max_i32 v18, v18, v35\n v_max_i32 v19, v19, v36\n v_max_i32 v20, v20, v37\n v_max_i32 v21, v21, v38\n v_max_i32 v22, v22, v39\n v_max_i32 v23, v23, v40\n ")
TEST(36, "v_min_f32 v16, v16, v33\n v_min_f32 v17, v17, v34\n v_min_f32 v18, v18, v35\n v_min_f32 v19, v19, v36\n v_min_f32 v20, v20, v37\n v_min_f32 v21, v21, v38\n v_min_f32 v22, v22, v39\n v_min_f32 v23, v23, v40\n ")
TEST(37, "v_fma_f32 v16, -v16, v33, v41\n v_fma_f32 v17, -v17, v34, v42\n v_fma_f32 v18, -v18, v35, v43\n v_fma_f32 v19, -v19, v36, v44\n v_fma_f32 v20, -v20, v37, v45\n v_fma_f32 v21, -v21, v38, v46\n v_fma_f32 v22, -v22, v39, v41\n v_fma_f32 v23, -v23, v40, v42\n ")
TEST(38, "v_sub_f32_e64 v16, |v16|, v33\n v_sub_f32_e64 v17, |v17|, v34\n v_sub_f32_e64 v18, |v18|, v35\n v_sub_f32_e64 v19, |v19|, v36\n v_sub_f32_e64 v20, |v20|, v37\n v_sub_f32_e64 v21, |v21|, v38\n v_sub_f32_e64 v22, |v22|, v39\n v_sub_f32_e64 v23, |v23|, v40\n ")
TEST(39, "v_bfe_u32 v16, v16, 4, 8\n v_bfe_u32 v17, v17, 4, 8\n v_bfe_u32 v18, v18, 4, 8\n v_bfe_u32 v19, v19, 4, 8\n v_bfe_u32 v20, v20, 4, 8\n v_bfe_u32 v21, v21, 4, 8\n v_bfe_u32 v22, v22, 4, 8\n v_bfe_u32 v23, v23, 4, 8\n ")
TEST2(40, "v_pk_mul_f32 v[16:17], v[16:17], v[32:33]\n v_pk_mul_f32 v[18:19], v[18:19], v[34:35]\n v_pk_mul_f32 v[20:21], v[20:21], v[36:37]\n v_pk_mul_f32 v[22:23], v[22:23], v[38:39]\n v_pk_mul_f32 v[24:25], v[24:25], v[40:41]\n v_pk_mul_f32 v[26:27], v[26:27], v[42:43]\n v_pk_mul_f32 v[28:29], v[28:29], v[44:45]\n v_pk_mul_f32 v[30:31], v[30:31], v[46:47]\n ", "v_mul_f32 v16, v16, v33\n v_mul_f32 v17, v17, v34\n v_mul_f32 v18, v18, v35\n v_mul_f32 v19, v19, v36\n v_mul_f32 v20, v20, v37\n v_mul_f32 v21, v21, v38\n v_mul_f32 v22, v22, v39\n v_mul_f32 v23, v23, v40\n ")
TEST(41, "v_fma_f64 v[16:17], v[16:17], v[32:33], v[32:33]\n v_fma_f64 v[18:19], v[18:19], v[34:35], v[34:35]\n v_fma_f64 v[20:21], v[20:21], v[36:37], v[36:37]\n v_fma_f64 v[22:23], v[22:23], v[38:39], v[38:39]\n v_fma_f64 v[24:25], v[24:25], v[40:41], v[40:41]\n v_fma_f64 v[26:27], v[26:27], v[42:43], v[42:43]\n v_fma_f64 v[28:29], v[28:29], v[44:45], v[44:45]\n v_fma_f64 v[30:31], v[30:31], v[46:47], v[46:47]\n ")
TEST(42, "v_mul_f64 v[16:17], v[16:17], v[32:33]\n v_mul_f64 v[18:19], v[18:19], v[34:35]\n v_mul_f64 v[20:21], v[20:21], v[36:37]\n v_mul_f64 v[22:23], v[22:23], v[38:39]\n v_mul_f64 v[24:25], v[24:25], v[40:41]\n v_mul_f64 v[26:27], v[26:27], v[42:43]\n v_mul_f64 v[28:29], v[28:29], v[44:45]\n v_mul_f64 v[30:31], v[30:31], v[46:47]\n ")
TEST(43, "v_cvt_f32_i32 v16, v33\n v_cvt_f32_i32 v17, v34\n v_cvt_f32_i32 v18, v35\n v_cvt_f32_i32 v19, v36\n v_cvt_f32_i32 v20, v37\n v_cvt_f32_i32 v21, v38\n v_cvt_f32_i32 v22, v39\n v_cvt_f32_i32 v23, v40\n ")
TEST(44, "v_readfirstlane_b32 s40, v33\n v_readfirstlane_b32 s40, v34\n v_readfirstlane_b32 s40, v35\n v_readfirstlane_b32 s40, v36\n v_readfirstlane_b32 s40, v37\n v_readfirstlane_b32 s40, v38\n v_readfirstlane_b32 s40, v39\n v_readfirstlane_b32 s40, v40\n ")
TEST(45, "v_cmp_class_f32 vcc, v16, v33\n v_cmp_class_f32 vcc, v17, v34\n v_cmp_class_f32 vcc, v18, v35\n v_cmp_class_f32 vcc, v19, v36\n v_cmp_class_f32 vcc, v20, v37\n v_cmp_class_f32 vcc, v21, v38\n v_cmp_class_f32 vcc, v22, v39\n v_cmp_class_f32 vcc, v23, v40\n ")
TEST(46, "v_ldexp_f32 v16, v16, v33\n v_ldexp_f32 v17, v17, v34\n v_ldexp_f32 v18, v18, v35\n v_ldexp_f32 v19, v19, v36\n v_ldexp_f32 v20, v20, v37\n v_ldexp_f32 v21, v21, v38\n v_ldexp_f32 v22, v22, v39\n v_ldexp_f32 v23, v23, v40\n ")
TEST(47, "v_frexp_mant_f32 v16, v33\n v_frexp_mant_f32 v17, v34\n v_frexp_mant_f32 v18, v35\n v_frexp_mant_f32 v19, v36\n v_frexp_mant_f32 v20, v37\n v_frexp_mant_f32 v21, v38\n v_frexp_mant_f32 v22, v39\n v_frexp_mant_f32 v23, v40\n ")
TEST(48, "v_mul_f32 v16, vcc_lo, v16\n v_mul_f32 v17, vcc_lo, v17\n v_mul_f32 v18, vcc_lo, v18\n v_mul_f32 v19, vcc_lo, v19\n v_mul_f32 v20, vcc_lo, v20\n v_mul_f32 v21, vcc_lo, v21\n v_mul_f32 v22, vcc_lo, v22\n v_mul_f32 v23, vcc_lo, v23\n ")
typedef void (*kfn)(float *);
int run(const char *name, kfn f, int waves_per_simd, int lanes_per_instr) {
  float *out; CHECK(hipMalloc(&out, 4));
  const int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, out);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 10; ++r) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, out);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double instr_per_simd = (double)waves_per_simd * 16.0 * ITERS;
  printf("%-44s waves/SIMD %d: %7.3f ms  %5.2f cycles per wave-instruction per SIMD (@2.4 GHz)\n", name, waves_per_simd, best,
         best * 1e-3 * 2.4e9 / instr_per_simd);
  (void)lanes_per_instr;
  CHECK(hipFree(out));
  return 0;
}

int main() {
  struct { const char *n; kfn f; } tests[] = {
      {"1 v_mul_f32 vgpr,vgpr (different banks)", t1}, {"2 v_mul_f32 vgpr,vgpr (same bank)", t2},
      {"3 v_mul_f32 sgpr,vgpr", t3}, {"4 v_mul_f32 inline,vgpr", t4}, {"5 v_mul_f32 literal,vgpr", t5},
      {"6 v_mul_f32_e64 vgpr,vgpr", t6}, {"7 v_fma_f32 3 vgpr", t7}, {"8 v_fma_f32 2 vgpr + sgpr", t8},
      {"9 v_pk_mul_f32 pair,pair", t9}, {"10 v_pk_mul_f32 pair,sgpr pair", t10}, {"11 v_pk_add_f32 pair,pair (+2 banks)", t11},
      {"12 v_pk_mul_f32 pair, vgpr broadcast", t12}, {"13 v_sub_f32 non-destructive", t13}, {"14 v_mov_b32", t14},
      {"15 v_pk_fma_f32 3 pairs", t15}, {"16 v_cndmask_b32 vcc", t16}, {"17 mix pk + 2 VOP2 (8 instr)", t17},
      {"18 v_mul_f32_e64 vgpr,sgpr", t18}, {"19 v_add_f32 sgpr,vgpr", t19}, {"20 v_cndmask_b32_e64 sgpr-pair mask", t20}, {"21 v_cmp_gt_f32 vcc (VOPC)", t21}, {"22 v_cmp_gt_f32_e64 sgpr pair", t22}, {"23 v_cvt_i32_f32", t23}, {"24 v_trunc_f32", t24}, {"25 v_rcp_f32", t25}, {"26 v_sqrt_f32", t26}, {"27 v_div_scale_f32", t27}, {"28 v_div_fmas_f32", t28}, {"29 v_div_fixup_f32", t29}, {"30 v_add_u32", t30}, {"31 v_lshlrev_b32 inline", t31}, {"32 v_and_b32 literal", t32}, {"33 v_mad_u32_u24", t33}, {"34 v_mov_b32_dpp row_shr:1", t34}, {"35 v_max_i32", t35}, {"36 v_min_f32", t36}, {"37 v_fma_f32 neg modifier", t37}, {"38 v_sub_f32 abs modifier (e64)", t38}, {"39 v_bfe_u32", t39}, {"40 phase mix: 8 pk then 8 VOP2", t40}, {"41 v_fma_f64", t41}, {"42 v_mul_f64", t42}, {"43 v_cvt_f32_i32", t43}, {"44 v_readfirstlane_b32", t44}, {"45 v_cmp_class_f32", t45}, {"46 v_ldexp_f32", t46}, {"47 v_frexp_mant_f32", t47}, {"48 v_mul_f32 vgpr, vcc_lo? (sgpr via vcc)", t48},
      {"49 wave mix: odd waves pk, even waves VOP2", t49}};
  for (int w : {4, 8})
    for (auto &t : tests)
      if (run(t.n, t.f, w, 64)) return 1;
  return 0;
}
