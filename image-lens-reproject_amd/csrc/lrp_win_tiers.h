// lrp_win_tiers.h — the tiers of the window kernel: how ONE pixel of a pass gets its bicubic sample out of the wavefront's
// LDS window, one function per tier.  All of them evaluate sample_bicubic + bicubicInterpolate of the reference
// (src/reproject.cpp:92-148) — four vertical cubics over the tap columns, then the horizontal one, in the reference's
// association order — and differ in what the window holds:
//
//   win_tier_coef      the raw window + three planes of weight-independent coefficients per tap-column origin
//                      (magnified mappings: 12 coefficient vectors + 4 taps per pixel, 6 instead of 17 operations per column)
//   win_tier_raw       the raw window only: 16 taps, five full cubics (1:1 mappings, split blocks, pass windows)
//   win_tier_edge_row  the block lies beyond the first / last source row: one source row + the plane of its vertical cubics
//   win_tier_edge_col  ... beyond the first / last source column: one source column
//
// `after_reads()` is called once, behind the pixel's last read of the window and ahead of its arithmetic: in the last pass of
// a block the caller requests the next window there (lrp_win_kernel.h: vmcnt retires in order, so the DMA must be younger
// than nothing but this pass's store).  RGBAZ (CH == 5): the depth channel comes from the float plane behind the colour slots.
#pragma once

#include "lrp_kernel_common.h"

namespace lrp {

// Depth of one pixel: its 16 taps from the float plane (`d` = the tap (int(sx) - 1, int(sy) - 1), `pitch` floats per row),
// the four vertical cubics as two packed ones (tap columns 0 | 1 and 2 | 3 in the halves of a register pair: each half of a
// v_pk_* rounds like the scalar instruction), then the horizontal one.
// Aligned: `d` and the rows behind it are 16-byte aligned (the taps of tap DMA: lrp_win_kernel.h request_taps) — a tap row is ONE
// ds_read_b128 instead of two ds_read2_b32.
template <bool Aligned = false, class After>
__device__ __forceinline__ float win_depth_sample(const float *d, int pitch, float fx, float fy, float hfx, float hfy, After after_reads) {
  float t[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if constexpr (Aligned) {
      const float4 v = *reinterpret_cast<const float4 *>(d + r * pitch);
      t[0][r] = v.x, t[1][r] = v.y, t[2][r] = v.z, t[3][r] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j][r] = d[r * pitch + j];
    }
  }
  after_reads();
  const f2 k01 = catmull_rom2(f2{t[0][0], t[1][0]}, f2{t[0][1], t[1][1]}, f2{t[0][2], t[1][2]}, f2{t[0][3], t[1][3]}, fy, hfy);
  const f2 k23 = catmull_rom2(f2{t[2][0], t[3][0]}, f2{t[2][1], t[3][1]}, f2{t[2][2], t[3][2]}, f2{t[2][3], t[3][3]}, fy, hfy);
  return catmull_rom(k01.x, k01.y, k23.x, k23.y, fx, hfx);
}

// Raw taps: `t` = slot of tap (int(sx) - 1, int(sy) - 1), `pitch` slots per window row; RGBAZ: `depth` = that tap in the
// float plane.  All 16 taps first (in the last pass they are the block's last reads of the window), then the cubics.
template <int CH, bool AlignedDepth = false, class After>
__device__ __forceinline__ Rgba win_tier_raw(const float4 *t, int pitch, const float *depth, float fx, float fy, After after_reads) {
  const float hfx = 0.5f * fx, hfy = 0.5f * fy;
  const float4 *t1 = t + pitch, *t2 = t1 + pitch, *t3 = t2 + pitch;
  Rgba q[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    q[j][0] = as_rgba(t[j]);
    q[j][1] = as_rgba(t1[j]);
    q[j][2] = as_rgba(t2[j]);
    q[j][3] = as_rgba(t3[j]);
  }
  if constexpr (CH != 5) after_reads();
  const Rgba k0 = cubic4(q[0][0], q[0][1], q[0][2], q[0][3], fy, hfy);
  const Rgba k1 = cubic4(q[1][0], q[1][1], q[1][2], q[1][3], fy, hfy);
  const Rgba k2 = cubic4(q[2][0], q[2][1], q[2][2], q[2][3], fy, hfy);
  const Rgba k3 = cubic4(q[3][0], q[3][1], q[3][2], q[3][3], fy, hfy);
  Rgba s = cubic4(k0, k1, k2, k3, fx, hfx);
  if constexpr (CH == 5) s.e = win_depth_sample<AlignedDepth>(depth, pitch, fx, fy, hfx, hfy, after_reads);
  return s;
}

// Coefficient tier.  Of the 17 operations of a vertical Catmull-Rom evaluation, 11 depend on the four taps only:
//     inner = ((3 (b - c)) + d) - a,  m0 = (((2 a - 5 b) + 4 c) - d),  cma = c - a
//     k = b + hfy * (cma + fy * (m0 + fy * inner))          (src/reproject.cpp:92-98)
// and were stored per tap-column origin in three planes behind the window (precompute() of the kernel).  `tb` = slot of tap
// (int(sx) - 1, int(sy)) — the second tap row, the only raw taps read —, `ci` = the pixel's first vector in plane 0, the planes
// `c_plane` slots apart.  Same operations on the same operands in the same order as the reference: bit for bit.
template <int CH, class After>
__device__ __forceinline__ Rgba win_tier_coef(const float4 *tb, const float4 *ci, int c_plane, int pitch, const float *depth, float fx, float fy,
                                              After after_reads) {
  const float hfx = 0.5f * fx, hfy = 0.5f * fy;
  const float4 *cm = ci + c_plane, *cc = cm + c_plane;
  // the only reads of the raw window: the second tap row
  const Rgba b0 = as_rgba(tb[0]), b1 = as_rgba(tb[1]), b2 = as_rgba(tb[2]), b3 = as_rgba(tb[3]);
  if constexpr (CH != 5) after_reads();
  // The four vertical evaluations step by step across the eight channel-pair chains instead of chain by chain: a step's
  // eight instructions are independent and the next step's operands are eight instructions old, so a wavefront never waits
  // on its own previous instruction (chain by chain, the compiler's order, every other instruction depends on its
  // predecessor and is preceded by an s_nop).  The plane reads are issued in the order the steps consume them.
  Rgba ci_[4], cm_[4], cc_[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ci_[j] = as_rgba(ci[j]);
#pragma unroll
  for (int j = 0; j < 4; ++j) cm_[j] = as_rgba(cm[j]);
#pragma unroll
  for (int j = 0; j < 4; ++j) cc_[j] = as_rgba(cc[j]);
  __builtin_amdgcn_sched_barrier(0);
  const Rgba bb[4] = {b0, b1, b2, b3};
  f2 t[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[2 * j] = fy * ci_[j].lo;
    t[2 * j + 1] = fy * ci_[j].hi;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[2 * j] = cm_[j].lo + t[2 * j];
    t[2 * j + 1] = cm_[j].hi + t[2 * j + 1];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = fy * t[i];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[2 * j] = cc_[j].lo + t[2 * j];
    t[2 * j + 1] = cc_[j].hi + t[2 * j + 1];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = hfy * t[i];
  __builtin_amdgcn_sched_barrier(0);
  Rgba kk[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    kk[j] = px_zero<4>();
    kk[j].lo = bb[j].lo + t[2 * j];
    kk[j].hi = bb[j].hi + t[2 * j + 1];
  }
  __builtin_amdgcn_sched_barrier(0);
  const Rgba k0 = kk[0], k1 = kk[1], k2 = kk[2], k3 = kk[3];
  Rgba s = cubic4(k0, k1, k2, k3, fx, hfx);
  if constexpr (CH == 5) s.e = win_depth_sample(depth, pitch, fx, fy, hfx, hfy, after_reads);
  return s;
}

// Beyond the first / last source row (src/reproject.cpp:114-147 with the four tap rows clamped to one): the four taps of a
// column are one texel t and its vertical cubic k = cubic(t, t, t, t, 0 or 1) comes from the plane behind the row (`kp` = the
// plane entry of tap column int(sx) - 1; RGBAZ: `kd` likewise in the float plane), leaving the horizontal cubic.
template <int CH, class After>
__device__ __forceinline__ Rgba win_tier_edge_row(const float4 *kp, const float *kd, float fx, After after_reads) {
  const float hfx = 0.5f * fx;
  const Rgba k0 = as_rgba(kp[0]), k1 = as_rgba(kp[1]), k2 = as_rgba(kp[2]), k3 = as_rgba(kp[3]);
  float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (CH == 5) {
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = kd[j];
  }
  after_reads();
  Rgba s = cubic4(k0, k1, k2, k3, fx, hfx);
  if constexpr (CH == 5) s.e = catmull_rom(d[0], d[1], d[2], d[3], fx, hfx);
  return s;
}

// Beyond the first / last source column: the four tap columns are one (`t` = its texel of row int(sy) - 1, consecutive slots
// are consecutive rows), their common vertical cubic K is evaluated once and the horizontal one is cubic(K, K, K, K, 0 or 1)
// with the clamped weight `fxc` (src/reproject.cpp:130).
template <int CH, class After>
__device__ __forceinline__ Rgba win_tier_edge_col(const float4 *t, const float *td, float fy, float fxc, After after_reads) {
  const float hfy = 0.5f * fy, hfxc = 0.5f * fxc;
  const Rgba t0 = as_rgba(t[0]), t1 = as_rgba(t[1]), t2 = as_rgba(t[2]), t3 = as_rgba(t[3]);
  float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (CH == 5) {
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = td[j];
  }
  after_reads();
  const Rgba kk = cubic4(t0, t1, t2, t3, fy, hfy);
  Rgba s = cubic4(kk, kk, kk, kk, fxc, hfxc);
  if constexpr (CH == 5) {
    const float kd = catmull_rom(d[0], d[1], d[2], d[3], fy, hfy);
    s.e = catmull_rom(kd, kd, kd, kd, fxc, hfxc);
  }
  return s;
}

} // namespace lrp
