// lrp_kernel_v2.h — the tile and window kernels: the hot path as it runs on gfx950.
//
// Work decomposition (tile kernel: nearest, bilinear, RGBAZ / super-sampled bicubic)
//   * A 256-thread workgroup (4 wavefronts) owns a 64 x 4R tile of output pixels
//     (R = 4 rows per wavefront for nearest, 2 for bilinear / bicubic).  Lane l of
//     every wavefront owns output column tile_x*64 + l; wavefront w owns the R
//     contiguous rows Rw .. Rw+R-1, so each of its stores is one contiguous 1 KiB
//     run (64 lanes x float4) and every per-row quantity is wave-uniform.
//   * Tiles are numbered so that the workgroups the dispatcher deals to one XCD
//     (blockIdx % 8 equal) walk bands of neighbouring tile rows (xcd_tile()): neighbouring
//     tiles read neighbouring source rows, which then hit in that XCD's 4 MiB L2, and
//     every XCD gets its share of every part of the frame.
//
// Instruction diet (the reference loop is VALU-bound on this chip, not HBM-bound: every
// packed, SGPR-operand, compare / select / convert / divide-helper instruction costs 4 issue
// cycles per wavefront, an IEEE divide 11 instructions, a sqrt ~12 — DESIGN.md section 5).
// Nothing below changes an operation, an operand or an order of the reference; work is only
// evaluated fewer times:
//   * separable output-lens terms come from per-column / per-row tables (lrp_tables.hip)
//     instead of 4 divides or 3 double-precision polynomials per pixel;
//   * lens-only constants (sensor_width / fov, angular spans) come from the host;
//   * x / -z, y / -z is skipped when -z == 1 for the whole wavefront (division by one is the
//     identity in IEEE arithmetic);
//   * the horizontal source coordinate of a rectilinear / equirectangular source comes from a
//     per-column table whenever the rotated ray's x and z do not depend on the row;
//   * without a rotation the four mirror images of a pixel share stage 1 of the coordinate
//     math (P.quad); for the equidistant target they share the ray under any rotation.
//
// Bicubic taps through LDS (window kernel, RGBA and RGB, one wavefront per workgroup)
//   The 16 taps of neighbouring pixels overlap almost completely, and 16 float4 gathers per
//   pixel saturate the texture-address path long before HBM.  A wavefront reduces the
//   tap-index bounding box of its 16 x 16 block with DPP-fused min / max, fetches that window
//   with LDS-DMA into its private 10 KiB of LDS (no barrier anywhere), evaluates the
//   weight-independent part of the vertical Catmull-Rom cubics once per window column and
//   reads 12 coefficient vectors + 4 taps per pixel at compile-time offsets from one address.
//   Windows larger than the LDS budget (poles, seam, strong minification) and non-consecutive
//   taps fall back, per wavefront, to explicit per-tap addressing.
#pragma once

#include <algorithm>

#include "lrp_device.h"
#include "lrp_source_axes.h"

#ifndef LRP_TILE_MINWAVES
#define LRP_TILE_MINWAVES 1 // __launch_bounds__ waves per SIMD of the tile kernel
#endif
#ifndef LRP_WIN_MINWAVES
#define LRP_WIN_MINWAVES 4 // __launch_bounds__ waves per SIMD of the window kernel (<= 128 VGPRs)
#endif
#ifndef LRP_TILE_ROWS_NN
#define LRP_TILE_ROWS_NN 4
#endif
#ifndef LRP_TILE_ROWS_BL
#define LRP_TILE_ROWS_BL 2
#endif
#ifndef LRP_TILE_ROWS_BC
#define LRP_TILE_ROWS_BC 2
#endif

namespace lrp {

#ifndef LRP_VERT_STEPS
#define LRP_VERT_STEPS 1 // coefficient tier: the vertical evaluations interleaved step by step (0: chain by chain, the compiler's order)
#endif
#ifndef LRP_OPT_POST
#define LRP_OPT_POST 1
#endif
#ifndef LRP_OPT_TIER
#define LRP_OPT_TIER 1
#endif
#ifndef LRP_WIN_ALIAS_PAIRS
#define LRP_WIN_ALIAS_PAIRS 1 // rectilinear -> panorama: the view and its copy behind the camera rendered side by side
#endif
constexpr int kT2W = 64;         // tile width: one output column per lane
// Output rows per wavefront of the tile kernel, per sampler (measured at the settled
// clock, 4K frames: bilinear and bicubic hold 4 / 16 taps per pixel in registers and run
// 5-15 % faster with 2 rows than with 4; nearest prefers 4).
template <int Interp> constexpr int tile_rows() {
  return Interp == 0 ? LRP_TILE_ROWS_NN : (Interp == 1 ? LRP_TILE_ROWS_BL : LRP_TILE_ROWS_BC);
}
constexpr int kT2Waves = 4;      // wavefronts per workgroup
constexpr int kT2Threads = 64 * kT2Waves;

// Batched launches: frame blockIdx.y of a batch of images that share one geometry (lenses, sizes,
// rotation): one launch keeps the wave slots full across frame boundaries — no inter-kernel gap,
// no drain of the last wavefronts before the next frame starts.
__device__ __forceinline__ KParams batch_frame(const KParams &Pk) {
  KParams P = Pk;
  if (Pk.batch_n > 0) {
    P.src = Pk.batch_src[blockIdx.y];
    P.dst = Pk.batch_dst[blockIdx.y];
  }
  return P;
}

// ---- wavefront-wide integer min / max (all 64 lanes active) ------------------
template <int Ctrl> __device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, Ctrl, 0xF, 0xF, false);
}
template <bool Max> __device__ __forceinline__ int pick(int a, int b) {
  if constexpr (Max)
    return a > b ? a : b;
  else
    return a < b ? a : b;
}
template <bool Max> __device__ __forceinline__ int wave_extreme(int v) {
  v = pick<Max>(v, dpp_i32<0xB1>(v));  // quad_perm [1,0,3,2]
  v = pick<Max>(v, dpp_i32<0x4E>(v));  // quad_perm [2,3,0,1]
  v = pick<Max>(v, dpp_i32<0x141>(v)); // row_half_mirror
  v = pick<Max>(v, dpp_i32<0x140>(v)); // row_mirror: every lane of a 16-lane row holds the row's extreme
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return pick<Max>(pick<Max>(a, b), pick<Max>(c, d));
}

// Wave-wide minima / maxima of six signed values (x range of a block, y ranges of its
// two halves) in 36 DPP-fused instructions: the butterfly runs inside v_min_i32 /
// v_max_i32 themselves (DPP on src0), the independent chains are interleaved (a VGPR
// written by a VALU instruction needs two wait states before a DPP read; five other
// instructions sit in between), and the last two steps fold the 16-lane rows with
// row_bcast:15 / row_bcast:31 so that lane 63 holds the result.
__device__ __forceinline__ void wave_box(int &lo_x, int &hi_x, int &lo_ya, int &hi_ya, int &lo_yb, int &hi_yb) {
#define LRP_BOX_STEP(CTRL)                                     \
  "v_min_i32_dpp %0, %0, %0 " CTRL "\n"                         \
  "v_max_i32_dpp %1, %1, %1 " CTRL "\n"                         \
  "v_min_i32_dpp %2, %2, %2 " CTRL "\n"                         \
  "v_max_i32_dpp %3, %3, %3 " CTRL "\n"                         \
  "v_min_i32_dpp %4, %4, %4 " CTRL "\n"                         \
  "v_max_i32_dpp %5, %5, %5 " CTRL "\n"
  asm volatile("s_nop 1\n" LRP_BOX_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
               LRP_BOX_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
               LRP_BOX_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
               LRP_BOX_STEP("row_mirror row_mask:0xf bank_mask:0xf")
               LRP_BOX_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
               LRP_BOX_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
               : "+v"(lo_x), "+v"(hi_x), "+v"(lo_ya), "+v"(hi_ya), "+v"(lo_yb), "+v"(hi_yb));
#undef LRP_BOX_STEP
  lo_x = __builtin_amdgcn_readlane(lo_x, 63);
  hi_x = __builtin_amdgcn_readlane(hi_x, 63);
  lo_ya = __builtin_amdgcn_readlane(lo_ya, 63);
  hi_ya = __builtin_amdgcn_readlane(hi_ya, 63);
  lo_yb = __builtin_amdgcn_readlane(lo_yb, 63);
  hi_yb = __builtin_amdgcn_readlane(hi_yb, 63);
}

__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(p) == ~0ull; }

// ---- ray -> source coordinates with hoisted constants --------------------------
template <int InMode>
__device__ __forceinline__ void ray_to_source_v2(const KParams &P, float x, float y, float z, float &cx, float &cy) {
  const LensP &L = P.in_lens;
  const float img_w = (float)P.in_w, img_h = (float)P.in_h;
  if constexpr (InMode == kInRect || InMode == kInEquidistant) {
    // x /= -z; y /= -z  (src/reproject.cpp:163-164,191-192).  v / 1.0f == v.
    const float nz = -z;
    if (!wave_all(nz == 1.0f)) {
      x = x / nz;
      y = y / nz;
    }
  }
  if constexpr (InMode == kInRect) {
    const float focal = L.p[0];
    cx = rect_axis(x, img_w, L.sensor_width, focal); // :165
    cy = rect_axis(y, img_h, L.sensor_height, focal);
  } else if constexpr (InMode == kInEquidistant) {
    const float r = lrp_sqrtf(x * x + y * y); // :193
    const float theta = atanf_(r);            // :194
    const float r_mm = P.in_focal * theta;    // :196-198
    const float r_px = r_mm / L.sensor_width * img_w;
    cx = x / r * r_px; // :202-203
    cy = y / r * r_px;
  } else {
    const float lat_min = L.p[0], lon_min = L.p[2];
    cx = equirect_cx(x, z, lon_min, P.in_lon_span, img_w);    // :262, :268
    cy = equirect_cy(x, y, z, lat_min, P.in_lat_span, img_h); // :263, :269
  }
}

// equidistant_to_vec (src/reproject.cpp:171-186) with the lens constant hoisted.
__device__ __forceinline__ void equidistant_ray_v2(const KParams &P, float cx, float cy, float &vx, float &vy,
                                                   float &vz) {
  const float r_px = lrp_sqrtf(cx * cx + cy * cy);
  const float r_mm = r_px / (float)P.out_w * P.out_lens.sensor_width;
  const float theta = r_mm / P.out_focal;
  float sn, cs;
  sincosf_(theta, sn, cs);
  const float s = sn / r_px;
  vx = s * cx;
  vy = s * cy;
  vz = cs;
}

// The reference's tap indices (src/reproject.cpp:114-127).
template <bool Loop>
__device__ __forceinline__ void bicubic_indices(float sx, float sy, int w, int h, int xs[4], int ys[4]) {
  xs[0] = column<Loop>(trunc_x86(sx - 1.0f), w);
  xs[1] = column<Loop>(trunc_x86(sx), w);
  xs[2] = column<Loop>(trunc_x86(sx + 1.0f), w);
  xs[3] = column<Loop>(trunc_x86(sx + 2.0f), w);
  ys[0] = clamp_index(trunc_x86(sy - 1.0f), h - 1);
  ys[1] = clamp_index(trunc_x86(sy), h - 1);
  ys[2] = clamp_index(trunc_x86(sy + 1.0f), h - 1);
  ys[3] = clamp_index(trunc_x86(sy + 2.0f), h - 1);
}

// ---- texels as channel pairs -------------------------------------------------------
// A texel of CH = 3, 4 or 5 channels is held as register pairs (c0,c1), (c2,c3)
// plus a single (c2 for RGB, c4 for RGBAZ); every interpolation step is one packed
// instruction per pair (v_pk_mul_f32 / v_pk_add_f32 round each half exactly like
// the scalar instruction) plus a scalar one for the odd channel, weights broadcast.
#ifndef LRP_NO_PACKED
#define LRP_NO_PACKED 0
#endif
#if LRP_NO_PACKED
// Channel pairs as two independent floats: every operation is a plain VOP2 / VOP3 instruction (build with
// -fno-slp-vectorize so that the compiler does not fuse them back into v_pk_*_f32).  On gfx950 a wavefront
// whose VALU stream contains packed-f32 instructions issues ALL its VALU instructions at ~4 cycles; a stream
// without them issues plain, SGPR-operand, convert and compare instructions at ~2.2 (tools/microbench/valu_runs.hip).
struct f2 {
  float x, y;
};
__device__ __forceinline__ f2 operator+(const f2 a, const f2 b) { return f2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f2 operator-(const f2 a, const f2 b) { return f2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f2 operator*(const f2 a, const f2 b) { return f2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f2 operator+(const f2 a, const float b) { return f2{a.x + b, a.y + b}; }
__device__ __forceinline__ f2 operator-(const f2 a, const float b) { return f2{a.x - b, a.y - b}; }
__device__ __forceinline__ f2 operator*(const f2 a, const float b) { return f2{a.x * b, a.y * b}; }
__device__ __forceinline__ f2 operator+(const float a, const f2 b) { return f2{a + b.x, a + b.y}; }
__device__ __forceinline__ f2 operator*(const float a, const f2 b) { return f2{a * b.x, a * b.y}; }
__device__ __forceinline__ f2 &operator+=(f2 &a, const f2 b) {
  a.x += b.x;
  a.y += b.y;
  return a;
}
__device__ __forceinline__ f2 &operator+=(f2 &a, const float b) {
  a.x += b;
  a.y += b;
  return a;
}
#else
typedef float f2 __attribute__((ext_vector_type(2)));
#endif
template <int CH> struct Px {
  f2 lo; // channels 0, 1
  f2 hi; // channels 2, 3 (CH >= 4)
  float e; // channel 2 (CH == 3) or 4 (CH == 5)
};
using Rgba = Px<4>;
__device__ __forceinline__ Rgba as_rgba(const float4 v) { return Rgba{f2{v.x, v.y}, f2{v.z, v.w}, 0.0f}; }

template <int CH> __device__ __forceinline__ Px<CH> px_zero() { return Px<CH>{f2{0.0f, 0.0f}, f2{0.0f, 0.0f}, 0.0f}; }
template <int CH> __device__ __forceinline__ void px_add(Px<CH> &a, const Px<CH> &b) {
  a.lo += b.lo;
  if constexpr (CH >= 4) a.hi += b.hi;
  if constexpr (CH & 1) a.e += b.e;
}

// cubicInterpolate (src/reproject.cpp:92-98), same association order as catmull_rom().
__device__ __forceinline__ f2 catmull_rom2(const f2 a, const f2 b, const f2 c, const f2 d, float t, float half_t) {
  const f2 inner = ((3.0f * (b - c)) + d) - a;
#if defined(LRP_FUSED_EXACT_PRODUCTS) && !LRP_NO_PACKED
  // Timing experiment (tools/ablate_units.sh; NOT the reference's bits for texels of 2^126 and more): 2 a and 4 c are exact
  // products unless they overflow, so fma(2, a, -(5 b)) and fma(4, c, x) round like the separate multiply + add — 15 instead of
  // 17 operations.  Guarding it needs a magnitude test of every texel, which costs what it saves.
  const f2 m5b = 5.0f * b;
  const f2 x0 = __builtin_elementwise_fma(f2{2.0f, 2.0f}, a, -m5b);
  const f2 x1 = __builtin_elementwise_fma(f2{4.0f, 4.0f}, c, x0);
  const f2 mid = (x1 - d) + t * inner;
#else
  const f2 mid = ((((2.0f * a) - (5.0f * b)) + (4.0f * c)) - d) + t * inner;
#endif
  const f2 outer = (c - a) + t * mid;
  return b + half_t * outer;
}
template <int CH>
__device__ __forceinline__ Px<CH> cubic_px(const Px<CH> &a, const Px<CH> &b, const Px<CH> &c, const Px<CH> &d, float t,
                                           float half_t) {
  Px<CH> r = px_zero<CH>();
  r.lo = catmull_rom2(a.lo, b.lo, c.lo, d.lo, t, half_t);
  if constexpr (CH >= 4) r.hi = catmull_rom2(a.hi, b.hi, c.hi, d.hi, t, half_t);
  if constexpr (CH & 1) r.e = catmull_rom(a.e, b.e, c.e, d.e, t, half_t);
  return r;
}
__device__ __forceinline__ Rgba cubic4(const Rgba a, const Rgba b, const Rgba c, const Rgba d, float t, float half_t) {
  return cubic_px<4>(a, b, c, d, t, half_t);
}

// ---- source texels through a buffer descriptor ------------------------------------
// buffer_load takes a 32-bit VGPR byte offset, an SGPR byte offset and a 12-bit
// immediate: the 16 taps of an interior bicubic pixel are ONE VGPR offset (first
// tap), four SGPR row offsets (0, pitch, 2 pitch, 3 pitch — computed once per
// kernel) and the immediates 0, T, 2T, 3T (T = texel bytes).  No per-tap address
// arithmetic at all.  RGB texels are one dwordx3, RGBAZ a dwordx4 + a dword.
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2))); // (a native vector: what the nontemporal builtins take)
typedef unsigned int u3 __attribute__((ext_vector_type(3)));
template <int CH>
__device__ __forceinline__ Px<CH> texel_at(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  if constexpr (CH == 3) {
    const u3 q = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)voff, (int)soff, 0);
    return Px<3>{f2{u2f(q.x), u2f(q.y)}, f2{0.0f, 0.0f}, u2f(q.z)};
  } else {
    const u4 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
    float e = 0.0f;
    if constexpr (CH == 5) e = u2f(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(voff + 16u), (int)soff, 0));
    return Px<CH>{f2{u2f(q.x), u2f(q.y)}, f2{u2f(q.z), u2f(q.w)}, e};
  }
}

// bicubicInterpolate (src/reproject.cpp:100-107): vertical cubic per tap column,
// then the horizontal one.  Taps: byte offset v[i] (column part, VGPR) + r[j]
// (row part; SGPR in the interior path).
template <int CH, bool ScalarRows, bool LowReg = false>
__device__ __forceinline__ Px<CH> bicubic_taps(__amdgpu_buffer_rsrc_t rsrc, uint32_t v0, uint32_t v1, uint32_t v2,
                                               uint32_t v3, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, float fx,
                                               float fy) {
  // ScalarRows: r[j] are wave-uniform and ride in the instruction's SGPR offset;
  // otherwise they are per-lane and are added into the VGPR offset.
  auto tap = [&](uint32_t v, uint32_t r) {
    if constexpr (ScalarRows)
      return texel_at<CH>(rsrc, v, r);
    else
      return texel_at<CH>(rsrc, v + r, 0u);
  };
  const float hfx = 0.5f * fx, hfy = 0.5f * fy;
  if constexpr (LowReg) {
    const Px<CH> k0 = cubic_px<CH>(tap(v0, r0), tap(v0, r1), tap(v0, r2), tap(v0, r3), fy, hfy);
    const Px<CH> k1 = cubic_px<CH>(tap(v1, r0), tap(v1, r1), tap(v1, r2), tap(v1, r3), fy, hfy);
    // the loads of columns 2, 3 stay behind the cubics of columns 0, 1: half the tap registers
    // live at once, one more memory round trip
    asm volatile("" ::: "memory");
    const Px<CH> k2 = cubic_px<CH>(tap(v2, r0), tap(v2, r1), tap(v2, r2), tap(v2, r3), fy, hfy);
    const Px<CH> k3 = cubic_px<CH>(tap(v3, r0), tap(v3, r1), tap(v3, r2), tap(v3, r3), fy, hfy);
    return cubic_px<CH>(k0, k1, k2, k3, fx, hfx);
  }
  // Loads in ROW-major order: the four taps of a row are 4 T contiguous bytes, i.e. one or two
  // 128-byte cache lines that the second to fourth load find in flight.  Column-major order
  // touches a row's line again only after 64 lanes x 4 rows of other lines have gone through
  // a 32 KiB L1 that 16 wavefronts share — under minification (every lane its own lines) that
  // fetches each line up to four times.
  const uint32_t v[4] = {v0, v1, v2, v3}, r[4] = {r0, r1, r2, r3};
  Px<CH> t[4][4];
  if constexpr (ScalarRows && (CH == 3 || CH == 5)) {
    // interior path (v1..v3 = v0 + T, 2T, 3T): the four texels of a tap row are 48 / 80
    // contiguous bytes — 3 / 5 dwordx4 loads instead of 4 dwordx3 / 4 dwordx4 + 4 dword
    constexpr int NV = CH == 3 ? 3 : 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f[NV * 4];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const u4 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(v0 + 16u * i), (int)r[j], 0);
        f[4 * i] = u2f(q.x);
        f[4 * i + 1] = u2f(q.y);
        f[4 * i + 2] = u2f(q.z);
        f[4 * i + 3] = u2f(q.w);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (CH == 3)
          t[c][j] = Px<3>{f2{f[3 * c], f[3 * c + 1]}, f2{0.0f, 0.0f}, f[3 * c + 2]};
        else
          t[c][j] = Px<CH>{f2{f[5 * c], f[5 * c + 1]}, f2{f[5 * c + 2], f[5 * c + 3]}, f[5 * c + 4]};
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i][j] = tap(v[i], r[j]);
  }
  if constexpr ((CH & 1) != 0) {
    // the single (third / fifth) channel of two tap columns shares a packed vertical cubic: same operations, each
    // half rounded like the scalar instruction; the paired channels go through cubic_px's packed path as before
    Px<CH> k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      k[i] = px_zero<CH>();
      k[i].lo = catmull_rom2(t[i][0].lo, t[i][1].lo, t[i][2].lo, t[i][3].lo, fy, hfy);
      if constexpr (CH >= 4) k[i].hi = catmull_rom2(t[i][0].hi, t[i][1].hi, t[i][2].hi, t[i][3].hi, fy, hfy);
    }
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
      const f2 e = catmull_rom2(f2{t[i][0].e, t[i + 1][0].e}, f2{t[i][1].e, t[i + 1][1].e}, f2{t[i][2].e, t[i + 1][2].e},
                                f2{t[i][3].e, t[i + 1][3].e}, fy, hfy);
      k[i].e = e.x;
      k[i + 1].e = e.y;
    }
    return cubic_px<CH>(k[0], k[1], k[2], k[3], fx, hfx);
  }
  const Px<CH> k0 = cubic_px<CH>(t[0][0], t[0][1], t[0][2], t[0][3], fy, hfy);
  const Px<CH> k1 = cubic_px<CH>(t[1][0], t[1][1], t[1][2], t[1][3], fy, hfy);
  const Px<CH> k2 = cubic_px<CH>(t[2][0], t[2][1], t[2][2], t[2][3], fy, hfy);
  const Px<CH> k3 = cubic_px<CH>(t[3][0], t[3][1], t[3][2], t[3][3], fy, hfy);
  return cubic_px<CH>(k0, k1, k2, k3, fx, hfx);
}

// Interior test shared by the bilinear and bicubic fast paths: with
// lo <= s < hi (hi = extent - reach) no tap index is clamped or wrapped, and with
// (s + reach) - s == reach the float additions s + 1.0f .. s + reach are exact, so
// int(s + k) == int(s) + k for every tap (s - 1.0f is exact for s >= 1).  NaN fails.
__device__ __forceinline__ int interior(float s, float lo, float hi, float reach) {
  return (int)(s >= lo) & (int)(s < hi) & (int)(((s + reach) - s) == reach);
}
// The exactness half of interior() is sufficient, not necessary: next to a power of two
// (2046 <= s < 2048 for reach 2) s + reach is rounded for half of all s, yet the truncation
// still lands on int(s) + reach unless s is within one ulp of the next integer.  The precise
// condition — asked only after the cheap vote has failed, i.e. for the stripe of blocks that
// crosses such a coordinate — is int(s + k) == int(s) + k for k = 1 .. reach, finite s.
__device__ __forceinline__ int taps_consecutive(float s, float reach) {
  const float t = __builtin_truncf(s);
  int ok = (int)(__builtin_truncf(s + 1.0f) == t + 1.0f);
  if (reach == 2.0f) ok &= (int)(__builtin_truncf(s + 2.0f) == t + 2.0f);
  return ok & (int)(__builtin_fabsf(s) < 8388608.0f);
}
__device__ __forceinline__ int interior_precise(float s, float lo, float hi, float reach) {
  return (int)(s >= lo) & (int)(s < hi) & taps_consecutive(s, reach);
}
// wave-wide: every lane interior (cheap test first)
__device__ __forceinline__ bool all_interior(float sx, float sy, float lo, float x_hi, float y_hi, float reach) {
  if (__builtin_amdgcn_ballot_w64((interior(sx, lo, x_hi, reach) & interior(sy, lo, y_hi, reach)) != 0) == ~0ull) return true;
  return __builtin_amdgcn_ballot_w64((interior_precise(sx, lo, x_hi, reach) & interior_precise(sy, lo, y_hi, reach)) != 0) == ~0ull;
}

// ---- output pixel -> source coordinates (src/reproject.cpp:287-324) ----------------
// Terms of the output lens that depend on the column and the horizontal
// sub-sample only.
struct ColTerms {
  float a, b; // rectilinear: vx | equirectangular: vx, vz | equidistant: scx
  // column-separable source x (P.xsep_tab, see lrp_tables.hip): the rotated ray's x and z
  // and the finished source texel x of this column
  float nx, nz, sx;
};
template <int OutLens> __device__ __forceinline__ ColTerms column_terms(const KParams &P, int xe, int ssx) {
  const int ns = P.num_samples;
  ColTerms c{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (OutLens != kEquidistant) {
    if (P.xsep_tab) {
      const int n = P.out_w * ns, j = xe * ns + ssx;
      c.nx = P.xsep_tab[j];
      c.nz = P.xsep_tab[n + j];
      c.sx = P.xsep_tab[2 * n + j];
    }
  }
  if constexpr (OutLens == kRect) {
    c.a = P.col_tab[xe * ns + ssx];
  } else if constexpr (OutLens == kEquirect) {
    c.a = P.col_tab[xe * ns + ssx];
    c.b = P.col_tab[P.out_w * ns + xe * ns + ssx];
  } else {
    const float cx = ((float)xe + 0.5f) - (float)P.out_w * 0.5f; // :287
    c.a = cx + ((float)ssx + 1.0f) / ((float)ns + 1.0f) - 0.5f;  // :295
  }
  return c;
}

// Stage 0: the ray of one sub-sample through the OUTPUT lens (src/reproject.cpp:152-158, 171-186,
// 245-257).  Mirroring the output pixel about the image centre negates vx / vy exactly.
template <int OutLens>
__device__ __forceinline__ void pixel_ray(const KParams &P, const ColTerms col, float row_v, int ye, int ssy, float &vx,
                                          float &vy, float &vz) {
  if constexpr (OutLens == kRect) {
    vx = col.a;
    vy = row_v;
    vz = -1.0f;
  } else if constexpr (OutLens == kEquirect) {
    vx = col.a;
    vz = col.b;
    vy = row_v;
  } else {
    const float cy = ((float)ye + 0.5f) - (float)P.out_h * 0.5f;                       // :288
    const float scy = cy + ((float)ssy + 1.0f) / ((float)P.num_samples + 1.0f) - 0.5f; // :298
    equidistant_ray_v2(P, col.a, scy, vx, vy, vz);
  }
}

// Rotation (:303-311) and projection through the INPUT lens up to the lens-plane coordinates.
template <int InMode>
__device__ __forceinline__ void ray_to_plane(const KParams &P, float vx, float vy, float vz, float &u, float &v) {
  if (P.has_rot) {
    const float nx = P.rot[0] * vx + P.rot[1] * vy + P.rot[2] * vz;
    const float ny = P.rot[3] * vx + P.rot[4] * vy + P.rot[5] * vz;
    const float nz = P.rot[6] * vx + P.rot[7] * vy + P.rot[8] * vz;
    vx = nx;
    vy = ny;
    vz = nz;
  }
  ray_to_source_v2<InMode>(P, vx, vy, vz, u, v);
}

// One sub-sample of output pixel (column terms `col`, row term `row_v` of row ye) ->
// top-left-origin source texel coordinates, in two stages.  All 64 lanes must be active
// (wave-wide vote inside).  row_v is unused for the equidistant target; ye / ssy are
// only used by it.
//
// Stage 1, pixel_plane(): everything up to the last quantity that changes only its sign
// when the output pixel is mirrored about the image centre (see the mirrored blocks of
// the window kernel):
//   rectilinear / equidistant source   (u, v) = lens-plane coordinates (px, py);
//   equirectangular source, xsep table  v = phi (latitude of the ray), u unused;
//   equirectangular source otherwise   (u, v) = (px, py), not mirrorable.
// Stage 2, plane_to_texel(): the rest of src/reproject.cpp:268-269 and :323-324.
template <int OutLens, int InMode>
__device__ __forceinline__ void pixel_plane(const KParams &P, const ColTerms col, float row_v, int ye, int ssy,
                                            float &u, float &v) {
  float vx, vy, vz;
  if constexpr (OutLens != kEquidistant && InMode != kInEquidistant) {
    // Column-separable source x: when the ray's x and z do not depend on the output row
    // (no rotation, or one whose [0][1] and [2][1] entries are zero) the source x of a
    // rectilinear / equirectangular source is a function of the column alone and comes
    // from a per-column table built with the very same operations; only y remains.
    if (P.xsep_tab) { // wave-uniform
      const float vz0 = OutLens == kRect ? -1.0f : col.b;
      float ny = row_v;
      if (P.has_rot) ny = P.rot[3] * col.a + P.rot[4] * row_v + P.rot[5] * vz0; // :308
      const LensP &L = P.in_lens;
      if constexpr (InMode == kInRect) {
        const float nz = -col.nz;
        if (!wave_all(nz == 1.0f)) ny = ny / nz; // :164
        v = rect_axis(ny, (float)P.in_h, L.sensor_height, L.p[0]);
      } else {
        v = equirect_phi(col.nx, ny, col.nz);
      }
      u = 0.0f;
      return;
    }
  }
  pixel_ray<OutLens>(P, col, row_v, ye, ssy, vx, vy, vz);
  ray_to_plane<InMode>(P, vx, vy, vz, u, v);
}

template <int OutLens, int InMode>
__device__ __forceinline__ void plane_to_texel(const KParams &P, const ColTerms col, float u, float v, float &sx,
                                               float &sy) {
  bool xsep = false;
  if constexpr (OutLens != kEquidistant && InMode != kInEquidistant) xsep = P.xsep_tab != nullptr;
  sx = xsep ? col.sx : texel_coord(u, (float)P.in_w); // :323
  if constexpr (InMode == kInEquirect || InMode == kInEquirectLoop) {
    if (xsep) v = equirect_cy_of_phi(v, P.in_lens.p[0], P.in_lat_span, (float)P.in_h); // :269
  }
  sy = texel_coord(v, (float)P.in_h); // :324
}

template <int OutLens, int InMode>
__device__ __forceinline__ void pixel_source_rt(const KParams &P, const ColTerms col, float row_v, int ye, int ssy,
                                                float &sx, float &sy) {
  float u, v;
  pixel_plane<OutLens, InMode>(P, col, row_v, ye, ssy, u, v);
  plane_to_texel<OutLens, InMode>(P, col, u, v, sx, sy);
}

// Row term of output row ye, sub-sample ssy (0 for the equidistant target, which has none).
template <int OutLens> __device__ __forceinline__ float row_term(const KParams &P, int ye, int ssy) {
  if constexpr (OutLens == kEquidistant)
    return 0.0f;
  else
    return P.row_tab[ye * P.num_samples + ssy];
}

template <int OutLens, int InMode>
__device__ __forceinline__ void pixel_source(const KParams &P, const ColTerms col, int ye, int ssy, float &sx,
                                             float &sy) {
  pixel_source_rt<OutLens, InMode>(P, col, row_term<OutLens>(P, ye, ssy), ye, ssy, sx, sy);
}

// ---- one sample, taps straight from global memory -------------------------------
struct SrcView {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t row_bytes;
  float x_hi, y_hi; // interior bounds of the fast paths: extent - reach
};
template <int Interp, int CH> __device__ __forceinline__ SrcView source_view(const KParams &P) {
  SrcView v;
  v.row_bytes = (uint32_t)P.in_w * (4u * CH);
  v.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.src), 0, (int)(v.row_bytes * (uint32_t)P.in_h),
                                             0x00020000);
  v.x_hi = (float)(P.in_w - (Interp == 2 ? 2 : 1));
  v.y_hi = (float)(P.in_h - (Interp == 2 ? 2 : 1));
  return v;
}

// Nearest / bilinear in two steps: sample_issue() selects the taps and requests them, sample_combine() is the
// arithmetic on the returned texels.  A caller that issues pixel p + 1 before it combines pixel p has two
// pixels' taps in flight per lane (the tile kernels below: with distinct sources resident the gathers come
// from HBM, and a wavefront that waits right behind its four loads exposes that latency once per pixel).
template <int Interp, int CH> struct TapSet;
template <int CH> struct TapSet<0, CH> {
  Px<CH> t;
};
template <int CH> struct TapSet<1, CH> {
  Px<CH> ll, lu, ul, uu;
  float fx, fy;
};
template <int Interp, bool Loop, int CH, int TexelBytes = 4 * CH>
__device__ __forceinline__ TapSet<Interp, CH> sample_issue(const KParams &P, const SrcView &src, float sx, float sy) {
  static_assert(Interp == 0 || Interp == 1, "nearest or bilinear");
  constexpr uint32_t T = (uint32_t)TexelBytes;
  const int in_w = P.in_w, in_h = P.in_h;
  const uint32_t row_bytes = src.row_bytes;
  const __amdgpu_buffer_rsrc_t rsrc = src.rsrc;
  TapSet<Interp, CH> taps;
  if constexpr (Interp == 1) {
    // sample_bilinear (src/reproject.cpp:55-90); the interior vote reduces the
    // indices to lx = int(sx), ux = lx + 1, fx = sx - lx
    uint32_t o_ll, o_lu, o_ul, o_uu;
    float fx, fy;
    if (all_interior(sx, sy, 0.0f, src.x_hi, src.y_hi, 1.0f)) {
      const float tx_ = __builtin_truncf(sx), ty_ = __builtin_truncf(sy);
      fx = sx - tx_;
      fy = sy - ty_;
      o_ll = __umul24((uint32_t)(int)ty_, row_bytes) + (uint32_t)(int)tx_ * T;
      o_lu = o_ll + T;
      o_ul = o_ll + row_bytes;
      o_uu = o_ul + T;
    } else {
      const int lx = column<Loop>(trunc_x86(sx), in_w), ux = column<Loop>(trunc_x86(sx + 1.0f), in_w);
      const int ly = clamp_index(trunc_x86(sy), in_h - 1), uy = clamp_index(trunc_x86(sy + 1.0f), in_h - 1);
      fx = unit_clamp(sx - (float)lx); // :70-71
      fy = unit_clamp(sy - (float)ly);
      const uint32_t rl = (uint32_t)ly * row_bytes, ru = (uint32_t)uy * row_bytes;
      o_ll = rl + (uint32_t)lx * T;
      o_lu = rl + (uint32_t)ux * T;
      o_ul = ru + (uint32_t)lx * T;
      o_uu = ru + (uint32_t)ux * T;
    }
    taps.fx = fx;
    taps.fy = fy;
    taps.ll = texel_at<CH>(rsrc, o_ll, 0u);
    taps.lu = texel_at<CH>(rsrc, o_lu, 0u);
    taps.ul = texel_at<CH>(rsrc, o_ul, 0u);
    taps.uu = texel_at<CH>(rsrc, o_uu, 0u);
  } else {
    // sample_nearest (src/reproject.cpp:39-53)
    const int lx = column<Loop>(trunc_x86(sx + 0.5f), in_w);
    const int ly = clamp_index(trunc_x86(sy + 0.5f), in_h - 1);
    taps.t = texel_at<CH>(rsrc, __umul24((uint32_t)ly, row_bytes) + (uint32_t)lx * T, 0u);
  }
  return taps;
}
template <int Interp, int CH> __device__ __forceinline__ Px<CH> sample_combine(const TapSet<Interp, CH> &taps) {
  if constexpr (Interp == 1) {
    const float fx = taps.fx, fy = taps.fy;
    const float cfx = 1.0f - fx, cfy = 1.0f - fy;
    Px<CH> s = px_zero<CH>();
    // l = fx*lu + cfx*ll; u = fx*uu + cfx*ul; r = fy*u + cfy*l  (:83-88)
    s.lo = fy * (fx * taps.uu.lo + cfx * taps.ul.lo) + cfy * (fx * taps.lu.lo + cfx * taps.ll.lo);
    if constexpr (CH >= 4) s.hi = fy * (fx * taps.uu.hi + cfx * taps.ul.hi) + cfy * (fx * taps.lu.hi + cfx * taps.ll.hi);
    if constexpr (CH & 1) s.e = fy * (fx * taps.uu.e + cfx * taps.ul.e) + cfy * (fx * taps.lu.e + cfx * taps.ll.e);
    return s;
  } else {
    return taps.t;
  }
}

// N pixels of one lane, their tap requests kept Depth pixels ahead of the arithmetic (bicubic: one pixel at a time).
// coords(p, sx, sy) is called once per pixel in increasing p, finish(p, sample) likewise.
// Measured (4096^2, 16 distinct sources per launch): nearest 74-77 -> 71-73 us RGBA (8 ahead), 94 -> 89 RGBAZ and
// 64 -> 57 RGB (all 16 ahead); bilinear is the same at 1, 2, 3 and 4 ahead (its registers cost occupancy) and stays at 1.
#ifndef LRP_TILE_DEPTH_BL
#define LRP_TILE_DEPTH_BL 1
#endif
#ifndef LRP_TILE_DEPTH_NN
#define LRP_TILE_DEPTH_NN (CH == 4 ? 8 : 16)
#endif
template <int Interp, bool Loop, int CH, int N, class Coords, class Finish>
__device__ __forceinline__ void sample_pixels(const KParams &P, const SrcView &src, Coords coords, Finish finish);

// sample_nearest / sample_bilinear / sample_bicubic (src/reproject.cpp:39-148).
// All 64 lanes must be active (wave-wide vote).
// TexelBytes != 4 * CH: the RGB window kernel's fallback reads its 12-byte texels as 16-byte
// vectors (CH = 4, TexelBytes = 12; dword alignment is all a buffer load needs and a read past the
// last texel returns 0) and discards the fourth component like the rest of that kernel.
template <int Interp, bool Loop, int CH, bool LowReg = false, int TexelBytes = 4 * CH>
__device__ __forceinline__ Px<CH> sample_direct(const KParams &P, const SrcView &src, float sx, float sy) {
  constexpr uint32_t T = (uint32_t)TexelBytes; // texel bytes
  const int in_w = P.in_w, in_h = P.in_h;
  const uint32_t row_bytes = src.row_bytes;
  const __amdgpu_buffer_rsrc_t rsrc = src.rsrc;
  const float x_hi = src.x_hi, y_hi = src.y_hi;
  Px<CH> s;
  if constexpr (Interp == 2) {
    if (all_interior(sx, sy, 1.0f, x_hi, y_hi, 2.0f)) {
      // every lane: 4 consecutive columns x 4 consecutive rows, nothing clamped
      // (src/reproject.cpp:114-131 reduce to int(s) - 1 .. int(s) + 2, f = s - int(s))
      const float tx_ = __builtin_truncf(sx), ty_ = __builtin_truncf(sy);
      const float fx = sx - tx_, fy = sy - ty_;
      uint32_t v0 = __umul24((uint32_t)((int)ty_ - 1), row_bytes) + (uint32_t)((int)tx_ - 1) * T;
#if defined(LRP_ABLATE_L2ROWS) // timing experiment (wrong results): the taps of every pixel come from the first 64 source rows (cache-resident), same access pattern within a row
      v0 = __umul24((uint32_t)(((int)ty_ - 1) & 63), row_bytes) + (uint32_t)((int)tx_ - 1) * T;
#endif
#if defined(LRP_ABLATE_ONETAP) // timing experiment (wrong results): one tap load per pixel, the arithmetic of all five cubics
      {
        const Px<CH> acc = texel_at<CH>(rsrc, v0, 0u);
        const float hfx_ = 0.5f * fx, hfy_ = 0.5f * fy;
        const Px<CH> k0 = cubic_px<CH>(acc, acc, acc, acc, fy, hfy_);
        Px<CH> k1 = k0, k2 = k0, k3 = k0;
        k1.lo += fx; k2.lo += fy; k3.lo += hfx_;
        s = cubic_px<CH>(cubic_px<CH>(k0, k1, k2, k3, fy, hfy_), cubic_px<CH>(k1, k2, k3, k0, fy, hfy_),
                         cubic_px<CH>(k2, k3, k0, k1, fy, hfy_), cubic_px<CH>(k3, k0, k1, k2, fy, hfy_), fx, hfx_);
      }
#else
      s = bicubic_taps<CH, true, LowReg>(rsrc, v0, v0 + T, v0 + 2u * T, v0 + 3u * T, 0u, row_bytes, 2u * row_bytes,
                                 3u * row_bytes, fx, fy);
#endif
    } else {
      int xs[4], ys[4];
      bicubic_indices<Loop>(sx, sy, in_w, in_h, xs, ys);
      const float fx = unit_clamp(sx - (float)xs[1]); // :130-131
      const float fy = unit_clamp(sy - (float)ys[1]);
      const float hfx = 0.5f * fx, hfy = 0.5f * fy;
      // Out-of-view pixels clamp all four tap columns (or rows) to one border index:
      // the four vertical cubics then have identical inputs, hence identical results,
      // and are evaluated once (4 or 1 loads instead of 16 scattered border gathers;
      // this is most of a rectilinear -> equirectangular frame).  Wave-uniform votes.
      const bool one_col = wave_all((xs[0] == xs[1]) & (xs[1] == xs[2]) & (xs[2] == xs[3]));
      const bool one_row = wave_all(ys[0] == ys[3]); // rows are clamped, never wrapped: monotone
      const uint32_t c0 = (uint32_t)xs[0] * T, r0 = (uint32_t)ys[0] * row_bytes;
      if (one_col && one_row) {
        const Px<CH> t = texel_at<CH>(rsrc, c0 + r0, 0u);
        const Px<CH> k = cubic_px<CH>(t, t, t, t, fy, hfy);
        s = cubic_px<CH>(k, k, k, k, fx, hfx);
      } else if (one_col) {
        const Px<CH> k = cubic_px<CH>(texel_at<CH>(rsrc, c0 + r0, 0u), texel_at<CH>(rsrc, c0 + (uint32_t)ys[1] * row_bytes, 0u),
                                      texel_at<CH>(rsrc, c0 + (uint32_t)ys[2] * row_bytes, 0u),
                                      texel_at<CH>(rsrc, c0 + (uint32_t)ys[3] * row_bytes, 0u), fy, hfy);
        s = cubic_px<CH>(k, k, k, k, fx, hfx);
      } else if (one_row) {
        const Px<CH> t0 = texel_at<CH>(rsrc, c0 + r0, 0u), t1 = texel_at<CH>(rsrc, (uint32_t)xs[1] * T + r0, 0u);
        const Px<CH> t2 = texel_at<CH>(rsrc, (uint32_t)xs[2] * T + r0, 0u), t3 = texel_at<CH>(rsrc, (uint32_t)xs[3] * T + r0, 0u);
        s = cubic_px<CH>(cubic_px<CH>(t0, t0, t0, t0, fy, hfy), cubic_px<CH>(t1, t1, t1, t1, fy, hfy),
                         cubic_px<CH>(t2, t2, t2, t2, fy, hfy), cubic_px<CH>(t3, t3, t3, t3, fy, hfy), fx, hfx);
      } else {
        s = bicubic_taps<CH, false, LowReg>(rsrc, c0, (uint32_t)xs[1] * T, (uint32_t)xs[2] * T, (uint32_t)xs[3] * T, r0,
                                    (uint32_t)ys[1] * row_bytes, (uint32_t)ys[2] * row_bytes,
                                    (uint32_t)ys[3] * row_bytes, fx, fy);
      }
    }
  } else {
    s = sample_combine<Interp, CH>(sample_issue<Interp, Loop, CH, TexelBytes>(P, src, sx, sy));
  }
  return s;
}


template <int Interp, bool Loop, int CH, int N, class Coords, class Finish>
__device__ __forceinline__ void sample_pixels(const KParams &P, const SrcView &src, Coords coords, Finish finish) {
  if constexpr (Interp == 2) {
#pragma unroll
    for (int p = 0; p < N; ++p) {
      float sx, sy;
      coords(p, sx, sy);
      finish(p, sample_direct<2, Loop, CH>(P, src, sx, sy));
    }
  } else {
    constexpr int kWant = Interp == 0 ? LRP_TILE_DEPTH_NN : LRP_TILE_DEPTH_BL;
    constexpr int D = kWant < 1 ? 1 : (kWant > N ? N : kWant);
    TapSet<Interp, CH> ring[D];
#pragma unroll
    for (int p = 0; p < D; ++p) {
      float sx, sy;
      coords(p, sx, sy);
      ring[p] = sample_issue<Interp, Loop, CH>(P, src, sx, sy);
    }
#pragma unroll
    for (int p = 0; p < N; ++p) {
      const Px<CH> sample = sample_combine<Interp, CH>(ring[p % D]); // waits for pixel p; pixels p + 1 .. p + D - 1 stay in flight
      if (p + D < N) {
        float sx, sy;
        coords(p + D, sx, sy);
        ring[p % D] = sample_issue<Interp, Loop, CH>(P, src, sx, sy);
      }
      finish(p, sample);
    }
  }
}

// src/reproject.cpp:338-341 + optional fused post_process (:421-437), one pixel.
// UnitNorm: num_samples == 1, normalize == 1.0f: x * 1.0f is x for every float (the sum 0.0f + s
// has already turned -0 into +0 and quieted a NaN), so the five multiplies are not issued.
template <int CH, bool UnitNorm = false>
__device__ __forceinline__ void finish_px(const KParams &P, const Px<CH> &a, float c[5]) {
  const float n = UnitNorm ? 1.0f : P.normalize;
  c[0] = a.lo.x;
  c[1] = a.lo.y;
  c[2] = CH == 3 ? a.e : a.hi.x;
  c[3] = a.hi.y;
  c[4] = a.e;
  if constexpr (!UnitNorm) {
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] *= n;
  }
  // (the flag is made opaque where it is tested: hoisted out of the pass loops as a lane mask, its negation for the
  // branch comes back as a v_cndmask + v_cmp pair in front of every store; as a scalar integer it is an s_cmp)
  int has_post = P.has_post;
#if LRP_OPT_POST
  asm volatile("" : "+s"(has_post));
#endif
  if (has_post != 0) {
    c[0] = tonemap(c[0], P.exposure, P.reinhard);
    c[1] = tonemap(c[1], P.exposure, P.reinhard);
    c[2] = tonemap(c[2], P.exposure, P.reinhard);
  }
}
// One finished pixel to `d`.  Non-temporal stores: the output is written once and never read by this kernel;
// keeping it out of the L2 leaves the cache to the source texels (measured on a 4K
// frame: nearest 88 -> 63 us, bilinear 117 -> 95 us, bicubic 223 -> 214 us).
template <int CH> __device__ __forceinline__ void store_texel_nt(float *d, const float c[5]) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
  typedef float v3f_a4 __attribute__((ext_vector_type(3), aligned(4)));
  if constexpr (CH == 4) {
    __builtin_nontemporal_store(v4f{c[0], c[1], c[2], c[3]}, reinterpret_cast<v4f *>(d));
  } else if constexpr (CH == 3) { // one dwordx3 per lane: a wavefront's row is 768 contiguous bytes
    __builtin_nontemporal_store(v3f_a4{c[0], c[1], c[2]}, reinterpret_cast<v3f_a4 *>(d));
  } else { // dwordx4 + dword (4-byte aligned): 1280 contiguous bytes per wavefront row
    __builtin_nontemporal_store(v4f_a4{c[0], c[1], c[2], c[3]}, reinterpret_cast<v4f_a4 *>(d));
    __builtin_nontemporal_store(c[4], d + 4);
  }
}
template <int CH, bool UnitNorm = false>
__device__ __forceinline__ void store_px(const KParams &P, uint32_t pixel_index, Px<CH> a) {
  float c[5];
  finish_px<CH, UnitNorm>(P, a, c);
  store_texel_nt<CH>(P.dst + (size_t)pixel_index * CH, c);
}
// RGBAZ output, a whole run of pixels per wavefront.  Stored per lane, a 20-byte pixel is a dwordx4 and a dword
// at a 20-byte lane stride: two instructions that each touch every 64-byte segment of the run and fill it only
// partly — measured at half the rate of whole segments (tools/microbench/store_stride.hip: 114 us against 62 us
// for the 335 MB of a 4096^2 frame).  So the wavefront's 64 pixels (1280 bytes: one row of 64 pixels, Rows == 1,
// or four rows of 16, Rows == 4, each row 320 contiguous bytes) are exchanged through 1.25 KiB of its own LDS
// (no barrier: LDS operations of one wavefront execute in order) and leave as 80 sixteen-byte chunks: lane i
// writes chunk i, lanes 0-15 chunks 64-79.  `slot`: this lane's pixel in run order (row * 16 + column for
// Rows == 4); `first`: pixel index of the run's first pixel; `row_step`: pixels from one run row to the next.
template <int Rows>
__device__ __forceinline__ void store_rgbaz_run(const KParams &P, float *lds, int slot, uint32_t first, int row_step,
                                                const float c[5]) {
  static_assert(Rows == 1 || Rows == 4, "one row of 64 pixels or four rows of 16");
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
  float *x = lds + slot * 5;
#pragma unroll
  for (int i = 0; i < 5; ++i) x[i] = c[i];
  const int lane = (int)(threadIdx.x & 63u);
  const v4f q0 = *reinterpret_cast<const v4f *>(lds + 4 * lane);
  float *const row0 = P.dst + (size_t)first * 5;
  if constexpr (Rows == 1) {
    __builtin_nontemporal_store(v4f_a4{q0.x, q0.y, q0.z, q0.w}, reinterpret_cast<v4f_a4 *>(row0 + 4 * lane));
    if (lane < 16) {
      const v4f q1 = *reinterpret_cast<const v4f *>(lds + 256 + 4 * lane);
      __builtin_nontemporal_store(v4f_a4{q1.x, q1.y, q1.z, q1.w}, reinterpret_cast<v4f_a4 *>(row0 + 256 + 4 * lane));
    }
  } else {
    const int r = (lane * 3277) >> 16, cc = lane - 20 * r; // chunk lane = chunk cc of run row r (20 chunks per row)
    const ptrdiff_t step = (ptrdiff_t)row_step * 5;
    __builtin_nontemporal_store(v4f_a4{q0.x, q0.y, q0.z, q0.w}, reinterpret_cast<v4f_a4 *>(row0 + r * step + 4 * cc));
    if (lane < 16) { // chunks 64-79: run row 3, chunks 4-19
      const v4f q1 = *reinterpret_cast<const v4f *>(lds + 256 + 4 * lane);
      __builtin_nontemporal_store(v4f_a4{q1.x, q1.y, q1.z, q1.w}, reinterpret_cast<v4f_a4 *>(row0 + 3 * step + 16 + 4 * lane));
    }
  }
}

// One row of a tile kernel's 64 pixels: RGBAZ rows that lie in the image whole (wave-uniform) go out as a run.
template <int CH, bool UnitNorm>
__device__ __forceinline__ void store_tile_row(const KParams &P, float *run_lds, bool whole_run, bool lane_inside, int lane_slot,
                                               uint32_t run_first, uint32_t pixel_index, const Px<CH> &a) {
  if constexpr (CH == 5) {
    if (whole_run) {
      float c[5];
      finish_px<5, UnitNorm>(P, a, c);
      store_rgbaz_run<1>(P, run_lds, lane_slot, run_first, 0, c);
      return;
    }
  }
  if (lane_inside) store_px<CH, UnitNorm>(P, pixel_index, a);
}

// ---- the tile kernel (RGB / RGBA / RGBAZ float) ----------------------------------
// Frames: the instantiation for batched launches whose wavefronts render their pixels for several consecutive frames of
// the batch (all frames share one geometry): the source coordinates of a wavefront's pixels are evaluated once, kept in
// registers, and every frame only requests its taps, interpolates and stores.
#ifndef LRP_TILE_MINWAVES_FRAMES
#define LRP_TILE_MINWAVES_FRAMES 4 // the frame-loop instantiations keep <= 128 VGPRs: they are bound by memory and need the wavefronts
#endif
// GeoRead: the instantiation whose pixels LOAD their source coordinates from a geometry-cache entry (lrp_geocache.h; the
// map is written as a side output by the plain path below when P.geo_mode == 1): nearest / bilinear, one sample per pixel,
// whole images; no lens math compiled in, the output lens is irrelevant (kRect by convention).
template <int OutLens, int InMode, int Interp, int CH, bool Frames = false, bool GeoRead = false>
__global__ __launch_bounds__(kT2Threads, (Frames || GeoRead) ? LRP_TILE_MINWAVES_FRAMES : LRP_TILE_MINWAVES) void reproject_tile_kernel(const KParams Pk) {
  constexpr bool Loop = (InMode == kInEquirectLoop);
  static_assert(!GeoRead || (!Frames && Interp != 2 && OutLens == kRect), "GeoRead tile kernel: nearest / bilinear, single launches");
  const int frames_per_wave = Frames ? (Pk.frames_per_wave > 0 ? Pk.frames_per_wave : 1) : 1;
  const int frame0 = Pk.batch_n > 0 ? (int)blockIdx.y * frames_per_wave : 0;
  const int n_frames = (Frames && Pk.batch_n > 0) ? min(frames_per_wave, Pk.batch_n - frame0) : 1;
  KParams P = Pk; // src / dst: the frame being rendered
  if (Pk.batch_n > 0) {
    P.src = Pk.batch_src[frame0];
    P.dst = Pk.batch_dst[frame0];
  }

  int tx, ty;
  if (!xcd_tile(P.tiles_x, P.tiles_y, tx, ty)) return; // whole workgroup
  // Alias pairs (see the window kernel): a rectilinear view rendered into a panorama appears a second time behind
  // the camera, from the same source texels; consecutive workgroups of an XCD take the two tiles that read them.
  if constexpr (LRP_WIN_ALIAS_PAIRS != 0 && (OutLens == kEquirect || GeoRead) && InMode == kInRect) {
    if (P.alias_pairs == 0) {
      // (a partial panorama has no second copy: raster order keeps neighbouring tiles together, 1-3 % faster there)
    } else if (P.quad == 1) { // quadrant tiles: columns from both ends inwards (tile t shares its texels with tile tiles_x-1-t)
      tx = (tx & 1) ? P.tiles_x - 1 - (tx >> 1) : (tx >> 1);
    } else if ((P.tiles_x & 1) == 0) { // tile (t, r) with tile (t + tiles_x/2, tiles_y-1-r)
      const bool second = (tx & 1) != 0;
      tx = (tx >> 1) + (second ? P.tiles_x >> 1 : 0);
      ty = second ? P.tiles_y - 1 - ty : ty;
    }
  }
  const int lane = (int)(threadIdx.x & 63u);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int x = tx * kT2W + lane;
  const int x0 = tx * kT2W; // first column of the tile (wave-uniform)
  float *run_lds = nullptr; // RGBAZ: the wavefront's exchange buffer (store_rgbaz_run)
  if constexpr (CH == 5) {
    __shared__ __attribute__((aligned(16))) float s_run[kT2Waves][320];
    run_lds = s_run[wave];
  }
  constexpr int kT2Rows = tile_rows<Interp>();
  const int y_first = P.y_offset + (ty * kT2Waves + wave) * kT2Rows; // wave-uniform
  // Lanes / rows beyond the image recompute the last valid pixel and never store
  // (all 64 lanes stay active for the wave-wide votes).
  SrcView src = source_view<Interp, CH>(P);
  // N pixels per lane: coordinates from coords(p, sx, sy), results to finish(p, sample) — for one frame straight through
  // sample_pixels, for several frames with the coordinates held in registers between the frames.
  auto render = [&](auto n_tag, auto coords, auto finish) {
    constexpr int N = decltype(n_tag)::value;
    if constexpr (!Frames) {
      sample_pixels<Interp, Loop, CH, N>(P, src, coords, finish);
    } else {
      float sxs[N], sys[N];
#pragma unroll
      for (int p = 0; p < N; ++p) {
        coords(p, sxs[p], sys[p]);
        __builtin_amdgcn_sched_barrier(0); // one pixel's coordinate math after the other (interleaved they need 200 registers)
      }
#pragma unroll 1
      for (int f = 0; f < n_frames; ++f) {
        P.src = Pk.batch_src[frame0 + f];
        P.dst = Pk.batch_dst[frame0 + f];
        src = source_view<Interp, CH>(P);
        sample_pixels<Interp, Loop, CH, N>(
            P, src,
            [&](int p, float &sx, float &sy) {
              sx = sxs[p];
              sy = sys[p];
            },
            finish);
      }
    }
  };
  if constexpr (GeoRead) {
    const int xg = x < P.out_w ? x : P.out_w - 1;
    const vf2 *const map = reinterpret_cast<const vf2 *>(P.geo_xy);
    auto coords = [&](int k, float &sx, float &sy) {
      const int yk = y_first + k;
      const vf2 v = __builtin_nontemporal_load(map + geo_map_index(xg, yk < P.y_end ? yk : P.y_end - 1, P.out_w));
      sx = v.x;
      sy = v.y;
    };
    auto finish = [&](int k, const Px<CH> &sample) {
      const int yk = y_first + k;
      Px<CH> a = px_zero<CH>();
      px_add<CH>(a, sample); // :334-336
      const bool row_inside = yk < P.y_end; // wave-uniform
      const uint32_t row_first = (uint32_t)yk * (uint32_t)P.out_w + (uint32_t)x0;
      store_tile_row<CH, true>(P, run_lds, row_inside && x0 + kT2W <= P.out_w, row_inside && x < P.out_w, lane, row_first,
                               row_first + (uint32_t)lane, a);
    };
    sample_pixels<Interp, Loop, CH, kT2Rows>(P, src, coords, finish);
    return;
  }
  if (P.quad == 2) {
    // Mirrored rays (equidistant target, num_samples == 1, any rotation): the ray through the
    // OUTPUT lens — a square root, sincosf and three divides per pixel that no table can hold,
    // the target is not separable — is odd in cx, cy (src/reproject.cpp:171-186: r_px is even,
    // vx = s * cx, vy = s * cy, vz = cos theta), so it is evaluated once per quadrant pixel; the
    // rotation and the source lens then run per mirror image as usual.
    if constexpr (OutLens == kEquidistant) {
      const int qw = (P.out_w + 1) >> 1, qh = (P.out_h + 1) >> 1;
      const int xq = x < qw ? x : qw - 1;
      const ColTerms col = column_terms<OutLens>(P, xq, 0);
      // pixel p of this lane: mirror image p & 3 of quadrant row p >> 2 (sample_pixels keeps the taps of the next
      // pixel(s) in flight while one is interpolated and stored)
      float vx = 0.0f, vy = 0.0f, vz = 0.0f;
      auto coords = [&](int p, float &sx, float &sy) {
        const int k = p >> 2, g = p & 3;
        const int yk = y_first + k;
        if (g == 0) pixel_ray<OutLens>(P, col, 0.0f, yk < qh ? yk : qh - 1, 0, vx, vy, vz); // (row: wave-uniform)
        const bool mx = (g & 1) != 0, my = (g >> 1) != 0;
        float u, v;
        // the centre column / row of an odd-sized image is its own mirror image: its ray component
        // is +0 and stays +0 (a -0 would be a different input to atan2f)
        const bool neg_x = mx && 2 * x != P.out_w - 1, neg_y = my && 2 * yk != P.out_h - 1;
        ray_to_plane<InMode>(P, neg_x ? -vx : vx, neg_y ? -vy : vy, vz, u, v);
        plane_to_texel<OutLens, InMode>(P, col, u, v, sx, sy);
      };
      auto finish = [&](int p, const Px<CH> &sample) {
        const int k = p >> 2, g = p & 3;
        const int yk = y_first + k;
        const bool mx = (g & 1) != 0, my = (g >> 1) != 0;
        Px<CH> a = px_zero<CH>();
        px_add<CH>(a, sample); // :334-336
        const int xo = mx ? P.out_w - 1 - x : x, yo = my ? P.out_h - 1 - yk : yk;
        store_tile_row<CH, true>(P, run_lds, x0 + kT2W <= qw && yk < qh, x < qw && yk < qh, mx ? kT2W - 1 - lane : lane,
                                 (uint32_t)yo * (uint32_t)P.out_w + (uint32_t)(mx ? P.out_w - x0 - kT2W : x0),
                                 (uint32_t)yo * (uint32_t)P.out_w + (uint32_t)xo, a);
      };
      // (one frame per wavefront: this path is bound by memory, and 8-16 pixels' coordinates in registers cost it occupancy:
      // equirect -> fisheye bilinear 124 -> 125-160 us with the frame loop)
      sample_pixels<Interp, Loop, CH, 4 * kT2Rows>(P, src, coords, finish);
    }
    return;
  }
  if (P.quad) {
    // Mirrored pixels (num_samples == 1, no rotation; see the window kernel below for why this is
    // exact): the launch enumerates the top-left quadrant, stage 1 of the coordinate math runs
    // once per quadrant pixel and serves its three mirror images as well.
    constexpr bool kInEqr = InMode == kInEquirect || InMode == kInEquirectLoop;
    const int qw = (P.out_w + 1) >> 1, qh = (P.out_h + 1) >> 1;
    const int xq = x < qw ? x : qw - 1;
    const ColTerms col = column_terms<OutLens>(P, xq, 0);
    const ColTerms col_m = column_terms<OutLens>(P, P.out_w - 1 - xq, 0);
    float qa = 0.0f, qb = 0.0f; // stage 1 of the current quadrant row
    auto coords = [&](int p, float &sx, float &sy) {
      const int k = p >> 2, g = p & 3;
      if (g == 0) {
        const int yk = y_first + k;
        const int yq = yk < qh ? yk : qh - 1; // wave-uniform
        float u, v;
        pixel_plane<OutLens, InMode>(P, col, row_term<OutLens>(P, yq, 0), yq, 0, u, v);
        if constexpr (kInEqr) { // through the column table (host guarantees it): v = phi; y texel for both signs
          float unused;
          plane_to_texel<OutLens, InMode>(P, col, u, v, unused, qa);
          plane_to_texel<OutLens, InMode>(P, col, u, -v, unused, qb);
        } else {
          qa = u;
          qb = v;
        }
      }
      const bool mx = (g & 1) != 0, my = (g >> 1) != 0;
      if constexpr (kInEqr) {
        sx = mx ? col_m.sx : col.sx;
        sy = my ? qb : qa;
      } else {
        plane_to_texel<OutLens, InMode>(P, mx ? col_m : col, mx ? -qa : qa, my ? -qb : qb, sx, sy);
      }
    };
    auto finish = [&](int p, const Px<CH> &sample) {
      const int k = p >> 2, g = p & 3;
      const int yk = y_first + k;
      const bool mx = (g & 1) != 0, my = (g >> 1) != 0;
      Px<CH> a = px_zero<CH>();
      px_add<CH>(a, sample); // :334-336
      const int xo = mx ? P.out_w - 1 - x : x, yo = my ? P.out_h - 1 - yk : yk;
      store_tile_row<CH, true>(P, run_lds, x0 + kT2W <= qw && yk < qh, x < qw && yk < qh, mx ? kT2W - 1 - lane : lane,
                               (uint32_t)yo * (uint32_t)P.out_w + (uint32_t)(mx ? P.out_w - x0 - kT2W : x0),
                               (uint32_t)yo * (uint32_t)P.out_w + (uint32_t)xo, a);
    };
    sample_pixels<Interp, Loop, CH, 4 * kT2Rows>(P, src, coords, finish); // (likewise one frame per wavefront)
    return;
  }
  const int xe = x < P.out_w ? x : P.out_w - 1;
  const int ns = P.num_samples;
  if (ns == 1 && Interp != 2) {
    // one sample per pixel, any rotation: the rows of this lane with their tap requests ahead of the arithmetic
    const ColTerms col = column_terms<OutLens>(P, xe, 0);
    const bool geo_write = !Frames && P.geo_mode == 1 && blockIdx.y == 0; // side output: the coordinate map of the geometry cache
    auto coords = [&](int k, float &sx, float &sy) {
      const int yk = y_first + k;
      const int ye = yk < P.y_end ? yk : P.y_end - 1; // (row: wave-uniform)
      pixel_source<OutLens, InMode>(P, col, ye, 0, sx, sy);
      if constexpr (!Frames)
        if (geo_write) reinterpret_cast<vf2 *>(P.geo_xy)[geo_map_index(xe, ye, P.out_w)] = vf2{sx, sy};
    };
    auto finish = [&](int k, const Px<CH> &sample) {
      const int yk = y_first + k;
      Px<CH> a = px_zero<CH>();
      px_add<CH>(a, sample); // :334-336
      const bool row_inside = yk < P.y_end; // wave-uniform
      const uint32_t row_first = (uint32_t)yk * (uint32_t)P.out_w + (uint32_t)x0;
      store_tile_row<CH, true>(P, run_lds, row_inside && x0 + kT2W <= P.out_w, row_inside && x < P.out_w, lane, row_first,
                               row_first + (uint32_t)lane, a);
    };
    render(std::integral_constant<int, kT2Rows>{}, coords, finish);
    return;
  }
  // (super-sampled pixels and the tile kernel's bicubic: one frame per wavefront; the host launches them that way)

  Px<CH> acc[kT2Rows];
#pragma unroll
  for (int k = 0; k < kT2Rows; ++k) acc[k] = px_zero<CH>();

  for (int ssx = 0; ssx < ns; ++ssx) {
    const ColTerms col = column_terms<OutLens>(P, xe, ssx);
    for (int ssy = 0; ssy < ns; ++ssy) {
#pragma unroll
      for (int k = 0; k < kT2Rows; ++k) {
        const int yk = y_first + k;
        const int ye = yk < P.y_end ? yk : P.y_end - 1; // wave-uniform
        float sx, sy;
        pixel_source<OutLens, InMode>(P, col, ye, ssy, sx, sy);
        const Px<CH> s = sample_direct<Interp, Loop, CH>(P, src, sx, sy);
        px_add<CH>(acc[k], s); // :334-336
      }
    }
  }

#pragma unroll
  for (int k = 0; k < kT2Rows; ++k) {
    const int yk = y_first + k;
    const bool row_inside = yk < P.y_end; // wave-uniform
    const uint32_t row_first = (uint32_t)yk * (uint32_t)P.out_w + (uint32_t)x0;
    if (ns == 1)
      store_tile_row<CH, true>(P, run_lds, row_inside && x0 + kT2W <= P.out_w, row_inside && x < P.out_w, lane, row_first,
                               row_first + (uint32_t)lane, acc[k]);
    else
      store_tile_row<CH, false>(P, run_lds, row_inside && x0 + kT2W <= P.out_w, row_inside && x < P.out_w, lane, row_first,
                                row_first + (uint32_t)lane, acc[k]);
  }
}

// ---- the bicubic window kernel (RGBA / RGB float) ------------------------------------
//
// 16 float4 gathers per pixel keep the texture-address path of a CU busy for ~256 cycles per
// wavefront and thrash its L1; the taps of neighbouring pixels overlap almost completely, so
// each wavefront stages the source window of its own 16 x 16 output block in LDS once:
//   * block = 16 x 16 output pixels per wavefront, 4 passes of 16 columns x 4 rows
//     (square blocks keep the window small under any rotation of the mapping);
//   * all 256 pixels interior (no clamped / wrapped tap; the common case) ->
//     window = [min int(sx) - 1, max int(sx) + 2] x [min int(sy) - 1, max int(sy) + 2],
//     reduced with DPP-fused v_min_i32 / v_max_i32, no LDS, no barrier;
//   * the window rows are fetched with global_load_lds_dwordx4 / dwordx3 (LDS-DMA: per-lane
//     global address, wave-uniform LDS row base + lane * 16; no VGPR round trip), lanes
//     beyond the window width masked off;
//   * s_waitcnt vmcnt orders the wavefront's own ds_reads behind its DMA — the window is
//     private to the wavefront, so there is no workgroup barrier at all;
//   * tier 1 (magnified mappings): the weight-independent 11 of the 17 operations of every
//     vertical cubic are evaluated once per window column and row into three coefficient
//     planes behind the window; a pixel reads 12 coefficient vectors + 4 taps;
//   * tier 2 (window fits, planes do not): a pixel's 16 taps are ONE LDS address + 3 row
//     increments and the immediates 0/16/32/48.
// A block with a border / seam / NaN pixel, or a window larger than the per-wave LDS budget
// (strong minification), takes sample_direct() per pass instead.
#ifndef LRP_WIN_CAP
#define LRP_WIN_CAP 640
#endif
#ifndef LRP_WIN_STRIP
#define LRP_WIN_STRIP 2
#endif
#ifndef LRP_WIN_COEF
#define LRP_WIN_COEF 1
#endif
static_assert(LRP_WIN_STRIP <= kGeoStripRows, "geometry-cache entries hold block rows in multiples of kGeoStripRows (lrp_params.h)");
constexpr bool kWinCoef = LRP_WIN_COEF != 0; // coefficient tier (below)
constexpr int kWinCap = LRP_WIN_CAP; // float4 texels per window buffer: 10 KiB per wavefront, 40 KiB per workgroup -> 4 workgroups / CU
#ifndef LRP_WIN_BLOCK_W
#define LRP_WIN_BLOCK_W 16
#endif
#ifndef LRP_WIN_WAVES
#define LRP_WIN_WAVES 1
#endif
// Wavefronts per workgroup of the window kernel.  Its wavefronts share nothing (the window is
// wave-private), so a workgroup is ONE wavefront: each of the 16 wave slots of a CU is refilled
// the moment its wavefront retires instead of when the slowest of four does.
constexpr int kWinWaves = LRP_WIN_WAVES;
constexpr int kWinThreads = 64 * kWinWaves;
constexpr int kBlkW = LRP_WIN_BLOCK_W;  // output block per wavefront: kBlkW x kBlkH = 256 pixels,
constexpr int kBlkH = 256 / kBlkW;      // 4 passes of kBlkW columns x (64 / kBlkW) rows
constexpr int kPassRows = 64 / kBlkW;

// Lane -> pixel of a pass (16 columns x 4 rows).  The LDS serves a ds_read_b128 in four groups
// of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md,
// LDS) — and only lanes of one group can conflict.  With LRP_WIN_LANEMAP each group renders ONE output row of
// the pass (quads of consecutive lanes stay four consecutive columns, so the stores are unchanged):
// the 16 pixels of a row read window slots that rise by 0 or 1 per pixel, i.e. distinct banks, and step to the next
// window row a few times at most.  A row-major mapping (lane = 16 row + column) puts half of two different rows into
// every group.  Measured twice — round 2, and round 3 with the frames of a batch sharing the coordinate math (LDS array
// 80 % busy with conflicts): bank-conflict cycles 151 M -> 90 M per 16-frame launch, LDS active cycles -12 %, frame time
// unchanged (101.7 vs 101.2 us).  Off by default.
#ifndef LRP_WIN_LANEMAP
#define LRP_WIN_LANEMAP 0
#endif
__device__ __forceinline__ void win_lane_pixel(int lane, int &prow, int &pcol) {
  static_assert(LRP_WIN_BLOCK_W == 16 || LRP_WIN_LANEMAP == 0, "the group lane map is written for 16-column passes");
  if constexpr (LRP_WIN_LANEMAP != 0) {
    const int m = lane & 31, seg = m >> 2;
    prow = ((lane >> 5) << 1) | ((0x96 >> seg) & 1);
    pcol = ((m >> 3) << 2) | (m & 3);
  } else {
    prow = lane / kBlkW;
    pcol = lane & (kBlkW - 1);
  }
}
#ifndef LRP_WIN_CORNER
#define LRP_WIN_CORNER 1 // blocks wholly beyond one corner of the source: one evaluation per block (0: per pixel)
#endif
#ifndef LRP_WIN_SPLIT
#define LRP_WIN_SPLIT 1 // blocks whose window exceeds the buffer but whose two half-block windows fit stage those one after the other (0: per-pixel gathers)
#endif
#ifndef LRP_WIN_PASSWIN
#define LRP_WIN_PASSWIN 1 // (with LRP_WIN_SPLIT) blocks whose half windows do not fit either try the window of each 16 x 4 pass
#endif
#ifndef LRP_OPAQUE_COL
#define LRP_OPAQUE_COL 1 // plain blocks: the column terms are opaque to loop-invariant code motion (see coords())
#endif
#ifndef LRP_WIN_EDGE
#define LRP_WIN_EDGE 1 // blocks wholly beyond one SIDE of the source (and inside it along the other axis): one source row / column staged (0: per-pixel gathers)
#endif
#ifndef LRP_WIN_STRIP_PLAN
#define LRP_WIN_STRIP_PLAN 1 // mirrored strips: one reduction for the windows of all four mirror blocks (0: one per block)
#endif

#if defined(LRP_TIER_STATS) // diagnostic builds (tools/ablate.sh): blocks per tier (coefficients, raw taps, direct)
__device__ unsigned g_tier_stats[8]; // coefficient, raw, direct, corner, beyond a row, beyond a column, split
#endif

// Source coordinates and window of one 16 x 16 block (4 pixels per lane).  Fat: the per-half plane offsets are
// stored (two more wave-uniform words per block) instead of re-derived with a few scalar instructions in every
// pass — the mirrored kernels have the SGPRs for that, the plain-block kernels, which also carry the next block's
// coordinates, do not.
template <bool Fat> struct WinBlockT {
  float sx[4], sy[4];
  int x_lo, y_lo, bw, bh, pitch; // window origin, size and row pitch in texels (wave-uniform)
  int y_lo2, bh2; // split blocks (tier bit 9): first row and height of the window of passes 2-3; y_lo / bh are those of passes 0-1
  __device__ __forceinline__ int spitch() const { return pitch; } // slot distance from window row r to r + 1
  __device__ __forceinline__ int org() const { return 0; }        // slot of window row 0
  // Wave-uniform state is kept small and integral: the kernel sits at the SGPR limit (every word held across the
  // block loop for `cur` and `nxt` pushes another one into a VGPR lane), and a bool that crosses the block loop
  // gets materialised through a VGPR (v_cndmask 0/1 + v_cmp) at every use.
  // tier: bit 0 staged (taps come from the LDS window), bit 1 coef (coefficient tier), bit 2 whole (one set of
  // planes for the block: iy0 / iyn equal for both halves), bits 3-5 corner (0, or 1 + (x beyond the right edge)
  // + 2 (y beyond the bottom edge): every pixel of the block lies beyond the same corner of the source — all its
  // taps clamp to that one corner texel with weights 0 / 1)
  int tier;
  __device__ __forceinline__ int staged() const { return tier & 1; }
  __device__ __forceinline__ int coef() const { return tier & 2; }
  __device__ __forceinline__ int whole() const { return tier & 4; }
  __device__ __forceinline__ int corner() const { return (tier >> 3) & 7; }
  // bits 6-8 edge (0, or 1 + (beyond the high side) + 2 (the side is in x)): every pixel of the block lies beyond the same
  // SIDE of the source along one axis and inside it along the other.  Beyond in y (1, 2): the four tap rows clamp to the
  // first / last source row and the vertical weight to 0 / 1, the window is bw texels of that row (bh = 1, pitch = bw)
  // followed by a plane of their vertical cubics.  Beyond in x (3, 4): the four tap columns clamp to the first / last
  // source column and the horizontal weight to 0 / 1, the window is bh texels of that column (bw = 1, pitch = 1).
  __device__ __forceinline__ int edge() const { return (tier >> 6) & 7; }
  // bit 9 split: the window of the whole block exceeds the buffer, the windows of its two halves (passes 0-1, 2-3) fit:
  // they are staged one after the other, raw taps (mappings around 1:1 whose blocks are a little too large: 60 % of the
  // blocks of an 8192^2 panorama -> 2048^2 cubemap face)
  __device__ __forceinline__ int split() const { return tier & 512; }
  __device__ __forceinline__ int rows_of(int half) const { return half ? bh2 : bh; }
  __device__ __forceinline__ int first_row_of(int half) const { return half ? y_lo2 : y_lo; }
  // coefficient tier: per half of the block (passes 0-1, 2-3) the first int(sy) and the number of distinct
  // int(sy) rows; a coefficient row has the window's pitch
  int iy0[2], iyn[2], c_plane, c_base; // plane size and first slot of plane 0 (behind the raw window + a margin)
  // slot of texel (int(sx) - 1, int(sy)) in the raw window = tap_base + int(sy) * spitch + int(sx)
  int tap_base;
  // ... and the slot of the pixel's first coefficient vector in plane 0 of half h lies this much further
  int c_delta_stored[2];
  __device__ __forceinline__ int c_delta_value(int h) const {
    const int c_org = c_base + (spitch() < 0 ? (iyn[h] - 1) * pitch : 0); // plane slot of the first origin row
    return c_org - iy0[h] * spitch() - (1 + x_lo) - tap_base;
  }
  __device__ __forceinline__ int c_delta(int h) const { return Fat ? c_delta_stored[h] : c_delta_value(h); }
};

// One wavefront walks its strip of `blocks_per_wave` blocks (plain: top to bottom; mirrored: a
// quadrant block and its three mirror images):
//     A(0); DMA(0)
//     for g:  A(g+1)                        | plain blocks: coordinates + box of the next block
//             s_waitcnt vmcnt(0 or 1)       | window g has landed
//             half 0: planes, passes 0, 1   | coefficient planes of the half, then its two passes
//             half 1: planes, passes 2, 3   | mirrored blocks derive A(g+1) at the start of pass 3
//                     DMA(g+1) inside pass 3, behind its last reads of the raw window and ahead
//                     of its arithmetic and its store (vmcnt retires in order)
// One 10 KiB buffer per wavefront: 4 wavefronts per SIMD (a double-buffered variant at 3 per
// SIMD measured 5-25 % slower).
//
// CH == 3 (RGB, what the PNG / JPEG path delivers): global_load_lds_dwordx3 reads 12 bytes per
// lane and writes them at a 16-byte lane stride (measured: the fourth dword of each slot is left
// untouched), i.e. the hardware expands RGB texels into RGBA-sized slots.  Everything after the
// DMA is therefore the RGBA code; the fourth component carries stale LDS contents through the
// arithmetic (no traps are enabled) and is never stored.
//
// QMode — which mirror images of a block one wavefront renders with a single evaluation of stage 1 of the
// coordinate math (everything between the output pixel and the last quantity that only changes sign under the mirror):
//   0  plain blocks: none (any rotation)
//   1  both axes (no rotation): the block and its three mirror images, g = 0..3, bit 0 mirrors x, bit 1 mirrors y
//   2  rows only: a rotation about the vertical axis (pan) leaves the mapping symmetric top / bottom — the rotated ray of
//      pixel (x, H-1-y) is the ray of (x, y) with its y negated, exactly (rows 0 and 2 of the matrix do not see vy: that
//      is the column-separable case, which the host requires; row 1 is (+-0, c, +-0)).  Images g = 0, 2.  The four
//      side faces of a cubemap, any --rotation pan,0,0.
//   3  columns only: a rotation about the horizontal axis (pitch) with a rectilinear target (vz = -1 exactly) leaves it
//      symmetric left / right: rows 1 and 2 of the matrix are (+-0, c, -s) / (+-0, s, c), so ny, nz do not see vx (the
//      zero products vanish in sums that end in the non-zero R5 vz, R8 vz) and nx = vx.  Images g = 0, 1.  The top and
//      bottom faces of a cubemap, any --rotation 0,pitch,0.
//   4  shared rays: an equidistant TARGET under any rotation.  Its ray costs a square root, a double-precision sincosf
//      and three divides per pixel, no table can hold it (the lens is not separable), and it is odd in cx and in cy by
//      construction (src/reproject.cpp:171-186: r_px is even, vx = s cx, vy = s cy, vz = cos theta) — so the ray is
//      evaluated once per quadrant pixel and its sign-flipped copies go through the rotation and the source lens per
//      mirror image, like the pixels of plain blocks.  Images g = 0..3.
// The host (lrp_capi.cpp win_mirror_mode) checks the matrix entries and the symmetry flags of the output-lens tables.
// Frames: the instantiation for batched launches whose wavefronts walk several frames (the frame loop costs the
// one-frame case registers, so single launches keep an instantiation without it).
// GeoRead: the instantiation that LOADS the source coordinates of its pixels and the window extremes of its blocks from
// a geometry-cache entry (lrp_params.h, lrp_geocache.h) instead of deriving them from the lenses: what the frames of a
// batch share in registers, single launches of one geometry share through HBM.  The entry is written as a side output by
// the plain-block instantiation (P.geo_mode == 1) the first time a geometry is rendered; the loaded values are the
// stored ones, so the rendered bits are the same.  No lens math is compiled in: the output lens is irrelevant (kRect by
// convention), plain blocks only.
template <int OutLens, int InMode, int QMode, int CH, bool Frames = false, bool GeoRead = false>
#ifndef LRP_WIN_MINWAVES5
#define LRP_WIN_MINWAVES5 3 // RGBAZ: 168 VGPRs (the 80 registers of a direct-path tap set do not fit 128 without spilling)
#endif
#ifndef LRP_WIN_MINWAVES_AXIS
#define LRP_WIN_MINWAVES_AXIS 4 // one-axis mirror modes (QMode 2, 3)
#endif
#ifndef LRP_WIN_MINWAVES_RAYS
#define LRP_WIN_MINWAVES_RAYS 3 // shared-ray mode (QMode 4): the rays of four pixels (12 VGPRs) next to the coordinates of two blocks do not fit 128; measured 251 us at four waves per SIMD (66 spilled registers), 227 at three, 236 plain
#endif
#ifndef LRP_WIN_MINWAVES_FRAMES
#define LRP_WIN_MINWAVES_FRAMES 4 // the instantiations with the frame loop
#endif
#ifndef LRP_WIN_CAP_BIG
#define LRP_WIN_CAP_BIG 1280 // window slots of the big-window GeoRead variant: 20 KiB per wavefront, two wavefronts per SIMD
#endif
__global__ __launch_bounds__(kWinThreads, (GeoRead && OutLens == kEquirect) ? 2 : CH == 5 ? LRP_WIN_MINWAVES5 : (QMode == 4 ? LRP_WIN_MINWAVES_RAYS : (Frames ? LRP_WIN_MINWAVES_FRAMES : (QMode >= 2 ? LRP_WIN_MINWAVES_AXIS : LRP_WIN_MINWAVES)))) void reproject_bicubic_win_kernel(const KParams Pk) {
  constexpr bool Quad = QMode != 0;
  constexpr bool MirX = QMode == 1 || QMode == 3 || QMode == 4, MirY = QMode == 1 || QMode == 2 || QMode == 4;
  constexpr bool kSharedRays = QMode == 4; // only the ray through the output lens is shared: per-image coordinates are stored like a plain block's
  static_assert(QMode != 4 || OutLens == kEquidistant, "shared rays: the equidistant target");
  constexpr int kAllMirrors = (MirX ? 1 : 0) | (MirY ? 2 : 0); // the image mirrored in every mirrored axis
  using WinBlock = WinBlockT<Quad>;
  static_assert(CH == 3 || CH == 4 || CH == 5, "window kernel: RGB, RGBA or RGBAZ");
  static_assert(QMode != 2 || (OutLens != kEquidistant && InMode != kInEquidistant), "rows-only mirroring goes through the column-separable source x");
  static_assert(QMode != 3 || OutLens == kRect, "columns-only mirroring needs vz == -1");
  static_assert(!GeoRead || (QMode == 0 && (OutLens == kRect || (OutLens == kEquirect && InMode == kInRect && !Frames))),
                "GeoRead: plain blocks, one instantiation per source mode (+ the big-window variant of the rectilinear source)");
  // The big-window variant (GeoRead, "OutLens" kEquirect by convention; chosen by the host for a rectilinear view rendered
  // into a panorama, BASELINE configs[3]): the in-view blocks of that mapping are minified 3-5 x 1.5-3 — the window of a 16 x 4
  // PASS is ~67 x 11 texels, too wide for one DMA instruction per row and too large for 10 KiB next to three other
  // wavefronts' — so this variant holds 20 KiB per wavefront (two wavefronts per SIMD, no register limit to speak of) and
  // stages pass windows up to 128 texels wide.  tools/microbench/row_gather.hip: rows of that shape arrive at 7.7 TB/s by
  // LDS-DMA with 8 wavefronts per CU, a window each in flight; per-pixel gathers of the same bytes at 4.5 TB/s in this kernel.
  constexpr bool kBigWin = GeoRead && OutLens == kEquirect;
  constexpr int kCap = kBigWin ? LRP_WIN_CAP_BIG : kWinCap; // 16-byte slots of this instantiation's window buffer
  constexpr int kMaxPassCols = kBigWin ? 128 : 64;          // widest pass window (texels): DMA instructions per window row = ceil(bw / 64)
  constexpr bool kGeoWrite = !GeoRead && !Frames && QMode == 0 && kWinWaves == 1; // (P.geo_mode == 1: the side output)
  const bool geo_write = kGeoWrite && (Pk.geo_mode == 1 || Pk.geo_mode == 3) && blockIdx.y == 0; // wave-uniform (3: the extremes only — the map is there; a batched launch: its first frame writes)
  // Frames of a batched launch share one geometry: the source coordinates of a pixel, the window of a block and its tier
  // are the same in every frame.  A wavefront therefore renders its strip for `frames_per_wave` consecutive frames
  // (blockIdx.y = group of frames) and runs everything that does not depend on the pixel DATA — stage 1 of the coordinate
  // math, the wave-wide box reductions, the window plan — once per block instead of once per block and frame.
  const int frames_per_wave = Frames ? (Pk.frames_per_wave > 0 ? Pk.frames_per_wave : 1) : 1;
  const int frame0 = Pk.batch_n > 0 ? (int)blockIdx.y * frames_per_wave : 0;
  const int n_frames = (Frames && Pk.batch_n > 0) ? min(frames_per_wave, Pk.batch_n - frame0) : 1;
  auto frame_src = [&](int f) { return Pk.batch_n > 0 ? Pk.batch_src[frame0 + f] : Pk.src; };
  auto frame_dst = [&](int f) { return Pk.batch_n > 0 ? Pk.batch_dst[frame0 + f] : Pk.dst; };
  KParams P = Pk; // src / dst: the frame being rendered (set_frame below)
  P.src = frame_src(0);
  P.dst = frame_dst(0);
  constexpr bool Loop = (InMode == kInEquirectLoop);
  // Edge blocks (WinBlockT::edge) are compiled for the rectilinear source only: a narrow view inside a wider target is
  // where whole blocks lie beyond one side of the source; in the other instantiations the extra code costs 2-3 % (measured:
  // fisheye -> rectilinear 100 -> 102.5 us, fisheye -> fisheye 141.5 -> 146) and such blocks take the per-pixel gathers.
  constexpr bool kEdge = LRP_WIN_EDGE != 0 && InMode == kInRect;
  // Split blocks (WinBlockT::split) are compiled into the single-launch instantiations only: in the kernels with the frame
  // loop the extra code costs 2-7 % on mappings that have no such block (measured: equirect -> rect 91 -> 98 us, rect ->
  // equirect 212 -> 227), and what needs them — the 2048^2 faces of an 8192^2 panorama — arrives as single launches.
  // ... and for panorama sources only (a large panorama rendered into smaller views is where blocks are a little too large;
  // the rectilinear-source kernels lost 6 % to the extra code: rect -> equirect 210 -> 223 us).
  // (... and for the RGBAZ kernels of a rectilinear source: their per-pixel path is 20 gathers a pixel, and the pass windows
  // below pay there — rect -> equirect RGBAZ 300 -> 288 us, BASELINE configs[3] — while RGBA / RGB lose 4-6 %.)
  constexpr bool kSplit = LRP_WIN_SPLIT != 0 && !Frames && (InMode == kInEquirect || InMode == kInEquirectLoop || (InMode == kInRect && CH == 5));
  // Pass windows (below) for rectilinear targets only — perspective views and cubemap faces out of a panorama; in the
  // fisheye-target kernels the extra code cost 2.5 % (equirect -> fisheye single launches 247 -> 253 us).
  constexpr bool kPassWin = (kSplit || (GeoRead && OutLens == kEquirect)) && LRP_WIN_PASSWIN != 0 && (OutLens == kRect || GeoRead || (InMode == kInRect && CH == 5));
  constexpr int kPlanes = 3;
  __shared__ float4 s_win[kWinWaves][kCap];

  int tx, ty;
  if (!xcd_tile<kWinXcdBand>(P.tiles_x, P.tiles_y, tx, ty)) return; // whole workgroup
  // Alias pairs (LRP_WIN_ALIAS_PAIRS; mirrored strips of rectilinear -> equirectangular).  The reference has no
  // hemisphere test: the ray of panorama pixel (x + W/2, H-1-y) is the ray of (x, y) with x and z negated, and a
  // rectilinear projection divides by z — both pixels land on (nearly: different roundings) the same source
  // texel, the view is rendered a second time behind the camera.  In quadrant terms the strip of tile column
  // t and the strip of column tiles_x-1-t read the same four source windows, mirror image g of the one being
  // image 3-g of the other.  Dealt in raster order the two are a quarter of a frame apart and the source is
  // fetched from HBM twice; here consecutive workgroups of an XCD take the columns from
  // both ends inwards (0, n-1, 1, n-2, ...) and the odd ones walk their mirror images in reverse, so the two
  // strips run side by side on one L2 and ask for the same lines at the same time.
  // Plain strips (the tables of a panorama are not mirror images bit for bit, so this is the kernel that runs):
  // the strip of tile (t, r) and the strip of tile (t + tiles_x/2, tiles_y-1-r) are the pair, the second one
  // walks its blocks bottom-up.
  constexpr bool kAliasPairs = LRP_WIN_ALIAS_PAIRS != 0 && (OutLens == kEquirect || GeoRead) && InMode == kInRect && kWinWaves == 1;
  int g_flip = 0;        // mirrored strips: the mirror image rendered by loop iteration g is g ^ g_flip
  bool g_reverse = false; // plain strips: iteration g renders block G-1-g
  if (kAliasPairs && P.alias_pairs != 0) {
    if constexpr (QMode == 1) {
      g_flip = (tx & 1) ? 3 : 0;
      tx = (tx & 1) ? P.tiles_x - 1 - (tx >> 1) : (tx >> 1);
    } else if constexpr (QMode == 2) {
      // rows-only strips span all columns: the partner of strip t is strip t + tiles_x/2, whose image 2 (bottom) reads what
      // image 0 (top) of this one reads
      if ((P.tiles_x & 1) == 0) {
        g_flip = (tx & 1) ? 2 : 0;
        tx = (tx >> 1) + ((tx & 1) ? P.tiles_x >> 1 : 0);
      }
    } else if ((P.tiles_x & 1) == 0) {
      g_reverse = (tx & 1) != 0;
      tx = (tx >> 1) + (g_reverse ? P.tiles_x >> 1 : 0);
      ty = g_reverse ? P.tiles_y - 1 - ty : ty;
    }
  }
  const int lane = (int)(threadIdx.x & 63u);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int Gs = P.blocks_per_wave; // blocks per strip of this launch
  // blocks this strip renders.  (A block wholly below the image re-renders the image's last row — every lane stores — which is
  // how the compute kernels keep their store count.  GeoRead skips such blocks: the entry holds extremes only for the block
  // rows the WRITING launch walked, and that launch may have cut its strips differently.)
  const int G = GeoRead ? min(Gs, (P.y_end - P.y_offset + kBlkH - 1) / kBlkH - ty * Gs) : Gs;
  auto block_row = [&](int g) { return (kAliasPairs && g_reverse) ? G - 1 - g : g; }; // block of a plain strip rendered by iteration g
  auto geo_block = [&](int g) { return (uint32_t)(ty * Gs + block_row(g)) * (uint32_t)P.tiles_x + (uint32_t)tx; }; // geometry cache: box record of a plain block
  // Mirrored blocks (Quad instantiations, launched when P.quad).  Without a rotation the mapping is symmetric
  // about both image axes: the pixels (x, y), (W-1-x, y), (x, H-1-y), (W-1-x, H-1-y) have
  // rays that differ in the signs of vx / vy only, every operation between the ray and the
  // lens-plane coordinates is an IEEE multiply, divide, square root of a sum of squares or
  // an odd libm function, so their plane coordinates differ in sign only — exactly.  The
  // launch then enumerates the top-left quadrant, a wavefront's "strip" is the block and
  // its three mirror images (g = 0..3: bit 0 mirrors x, bit 1 mirrors y), and stage 1 of
  // the coordinate math (pixel_plane) runs once for the four of them.
  constexpr bool quad = Quad;
  const int qw = MirX ? (P.out_w + 1) >> 1 : P.out_w; // columns / rows enumerated by the launch
  const int qh = MirY ? (P.out_h + 1) >> 1 : P.y_end; // (a row band: rows beyond it re-render its last row)
  // mirror image rendered by loop iteration g of a mirrored strip (bit 0: mirrored in x, bit 1: in y)
  // (masked: the compiler then knows that an axis which is not mirrored never selects the mirrored column / sign)
  auto image_of = [&](int g) { return ((QMode == 2 ? 2 * g : g) ^ g_flip) & (QMode == 2 ? kAllMirrors : -1); };
  // workgroup tile = 16 kWinWaves x 16G (x 16 of the quadrant when mirrored): one strip per wavefront
  int prow, pcol; // this lane's pixel of a pass
  win_lane_pixel(lane, prow, pcol);
  const int x = tx * (kBlkW * kWinWaves) + wave * kBlkW + pcol;
  const int y_lane = P.y_offset + ty * (quad ? kBlkH : kBlkH * Gs) + prow; // + kBlkH * g + kPassRows * pass
  const int xe = x < qw ? x : qw - 1;
  const int in_w = P.in_w;
  SrcView src = source_view<2, CH>(P);
  auto set_frame = [&](int f) {
    P.src = frame_src(f);
    P.dst = frame_dst(f);
    src = source_view<2, CH>(P);
  };
  float4 *const win0 = s_win[wave];
  constexpr bool kRunsEverywhere = CH == 5 && (OutLens == kEquirect || GeoRead) && InMode == kInRect; // (GeoRead: and P.rgbaz_runs)
  float *out_lds = nullptr; // RGBAZ: the wavefront's exchange buffer of store_rgbaz_run (three waves per SIMD: the LDS is there)
  if constexpr (CH == 5) {
    __shared__ __attribute__((aligned(16))) float s_out[kWinWaves][320];
    out_lds = s_out[wave];
  }
  ColTerms col{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (!GeoRead) col = column_terms<OutLens>(P, xe, 0);
  ColTerms col_m = col; // the mirrored column
  if constexpr (MirX && !kSharedRays) col_m = column_terms<OutLens>(P, P.out_w - 1 - xe, 0);
  // Stage-1 results of the four quadrant pixels of this lane, kept for the whole strip:
  //   rectilinear / equidistant source: (qa, qb) = plane coordinates (u, v); a mirror image negates them;
  //   equirectangular source (through the xsep table): qa, qb = source texel y for +phi and for -phi
  //   (the division of :269 once per sign, not once per mirror image); x comes from the column tables.
  //   columns-only mirroring of an equirectangular source (no column table: the rotation pitches): the longitude
  //   theta = -atan2f(-nx, -nz) is odd in nx (lrp_math.h atan2f_: the sign of y only selects +-z, tests/test_math_vs_libm.py),
  //   the latitude does not see its sign: qa, qc = source texel x for +theta and for -theta, qb = source texel y.
  float qa[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  constexpr bool kInEqr = InMode == kInEquirect || InMode == kInEquirectLoop;
  constexpr bool kEqrByTheta = QMode == 3 && kInEqr;
  //   shared rays: (qa, qb, qc) = the ray (vx, vy, vz) of the quadrant pixel.
  float qc[(kEqrByTheta || kSharedRays) ? 4 : 1] = {};
  auto quad_xy = [&](int g, int k, float &sx, float &sy) { // source texel coordinates of pixel k of mirror image g
    const bool mx = (g & 1) != 0, my = (g >> 1) != 0;
    if constexpr (kSharedRays) {
      // the mirrored pixel's ray: vx / vy negated — except in the centre column / row of an odd-sized image, which is
      // its own mirror image: its component is the +0 of s * 0 and stays +0 (a -0 is another input to atan2f)
      const int yk = y_lane + kPassRows * k;
      const uint32_t sgn_x = (mx && 2 * xe != P.out_w - 1) ? 0x80000000u : 0u;
      const uint32_t sgn_y = (my && 2 * yk != P.out_h - 1) ? 0x80000000u : 0u;
      float u, v;
      ray_to_plane<InMode>(P, u2f(f2u(qa[k]) ^ sgn_x), u2f(f2u(qb[k]) ^ sgn_y), qc[k], u, v);
      plane_to_texel<OutLens, InMode>(P, col, u, v, sx, sy);
    } else if constexpr (kEqrByTheta) {
      sx = mx ? qc[k] : qa[k];
      sy = qb[k];
    } else if constexpr (kInEqr) {
      sx = mx ? col_m.sx : col.sx;
      sy = my ? qb[k] : qa[k];
    } else {
      // exact negation = the sign bit flipped by a wave-uniform mask: one v_xor with an SGPR operand, no
      // second register holding -q next to q (a select between the two costs 8 VGPRs, which spilled)
      const uint32_t sgn_x = mx ? 0x80000000u : 0u, sgn_y = my ? 0x80000000u : 0u;
      plane_to_texel<OutLens, InMode>(P, mx ? col_m : col, u2f(f2u(qa[k]) ^ sgn_x), u2f(f2u(qb[k]) ^ sgn_y), sx, sy);
    }
  };

  // 16-byte LDS slots of the raw window
  // RGBAZ (CH == 5): the window is two planes of the same pitch x bh geometry — colour (one 16-byte slot
  // per texel, fetched with global_load_lds_dwordx4 from the texel's first four floats, 20-byte texel
  // stride) and, right behind it, depth (one float per texel, global_load_lds_dword from its fifth).
  // Everything written for RGBA then serves the colour channels unchanged; depth reads its 16 taps from
  // the float plane and runs the five cubics as scalar instructions.
  // the thresholds of plan_window as float bits in scalar registers (an int -> float conversion is a vector instruction:
  // left to the compiler its result stays in a vector register for the whole kernel, and spills)
  const int x_hi_bits = __builtin_amdgcn_readfirstlane((int)f2u((float)(P.in_w - 2)));
  const int y_hi_bits = __builtin_amdgcn_readfirstlane((int)f2u((float)(P.in_h - 2)));
  const int beyond_x_bits = __builtin_amdgcn_readfirstlane((int)f2u((float)(P.in_w + 1)));
  const int beyond_y_bits = __builtin_amdgcn_readfirstlane((int)f2u((float)(P.in_h + 1)));
  auto slots_of_rows = [](int pitch, int rows) { return CH == 5 ? pitch * rows + ((pitch * rows + 3) >> 2) : pitch * rows; };
  auto raw_slots = [&](const WinBlock &b) { return slots_of_rows(b.pitch, kSplit ? max(b.bh, b.bh2) : b.bh); }; // (bh2 == 0 unless split)
  // Window of a block from the wave-wide extremes of its source coordinates (float bits, see below):
  // x range of the block, y ranges of its two halves.
  // exact_x / exact_y: int(s + k) == int(s) + k, k = -1 .. 2, holds for every pixel's x / y (the exactness vote of coords())
  auto plan_window = [&](WinBlock &b, int w_lo_x, int w_hi_x, int w_lo_ya, int w_hi_ya, int w_lo_yb, int w_hi_yb, bool exact_x,
                         bool exact_y) {
    const int w_lo_y = min(w_lo_ya, w_lo_yb), w_hi_y = max(w_hi_ya, w_hi_yb);
    const int one = (int)f2u(1.0f);
    // 1 <= s < extent - 2 for every pixel: every tap index is int(s) - 1 .. int(s) + 2, unclamped
    const bool in_x = exact_x && w_lo_x >= one && w_hi_x < x_hi_bits;
    const bool in_y = exact_y && w_lo_y >= one && w_hi_y < y_hi_bits;
    if (in_x && in_y) {
      // float -> int of the wave-uniform extremes (VALU has the converter)
      const int x_first = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_lo_x));
      const int x_last = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_hi_x));
      const int ya_first = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_lo_ya));
      const int ya_last = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_hi_ya));
      const int yb_first = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_lo_yb));
      const int yb_last = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_hi_yb));
      const int y_first = min(ya_first, yb_first), y_last = max(ya_last, yb_last);
      b.x_lo = x_first - 1;
      b.y_lo = y_first - 1;
      b.bw = x_last + 2 - b.x_lo + 1;
      b.bh = y_last + 2 - b.y_lo + 1;
      b.pitch = b.bw | 1; // odd: consecutive window rows start an odd number of 16 B slots apart
      b.tier = (b.bw <= 64 && raw_slots(b) <= kCap) ? 1 : 0;
      if (kSplit && P.win_split != 0 && b.tier == 0 && b.bw <= 64) {
        const int a_lo = ya_first - 1, a_rows = ya_last + 2 - a_lo + 1, b_lo = yb_first - 1, b_rows = yb_last + 2 - b_lo + 1;
        if (slots_of_rows(b.pitch, max(a_rows, b_rows)) <= kCap) {
          b.y_lo = a_lo;
          b.bh = a_rows;
          b.y_lo2 = b_lo;
          b.bh2 = b_rows;
          b.tier = 1 | 512;
        }
      }
      // coefficient tier: three planes of pitch x iyn[h] tap-column origins behind the raw window
      b.iy0[0] = ya_first;
      b.iyn[0] = ya_last - ya_first + 1;
      b.iy0[1] = yb_first;
      b.iyn[1] = yb_last - yb_first + 1;
      // strongly magnified blocks have room for the planes of ALL their origin rows: one
      // precompute per block (fuller lanes: e.g. 132 origins in 3 trips instead of 2 x 77 in 4)
      if (!b.split() && raw_slots(b) + kPlanes * b.pitch * (y_last - y_first + 1) <= kCap) b.tier |= 4;
      if (b.whole()) {
        b.iy0[0] = b.iy0[1] = y_first;
        b.iyn[0] = b.iyn[1] = y_last - y_first + 1;
      }
      b.c_plane = b.pitch * max(b.iyn[0], b.iyn[1]);
      if (kWinCoef && P.win_coef != 0 && b.staged() && !b.split() && raw_slots(b) + kPlanes * b.c_plane <= kCap) b.tier |= 2;
      // planes behind the raw window plus, where there is room, one row and one column of slack:
      // the next block's (slightly different) window can then be requested while this block's
      // planes are still being read (see next_window)
      b.c_base = min(raw_slots(b) + b.pitch + b.bh + 1, kCap - kPlanes * b.c_plane);
      b.tap_base = b.org() - b.y_lo * b.spitch() - (1 + b.x_lo);
      if constexpr (Quad) {
        b.c_delta_stored[0] = b.c_delta_value(0);
        b.c_delta_stored[1] = b.c_delta_value(1);
      }
    } else if constexpr (!Loop) {
      // Out of view beyond one CORNER of the source (most of a narrow view inside a panorama: 37 % of the blocks of
      // rectilinear -> equirectangular).  s <= -2: the four tap indices int(s - 1) .. int(s + 2) clamp to 0 and the
      // weight clamp(s - 0, 0, 1) is 0; extent + 1 <= s < 2^31: they clamp to extent - 1, the weight is 1
      // (src/reproject.cpp:114-131; from 2^31 on cvttss2si gives INT_MIN and the index clamps to 0 instead).  Then
      // every pixel of the block is the same function of the same corner texel: evaluated once per block.
      // On the raw bits: negative floats order backwards as signed integers, -2.0 .. -inf is 0xC0000000 .. 0xFF800000
      // (a negative NaN lies above that range, a positive one above 2^31).
      auto side = [](int lo, int hi, int beyond_bits) { // 0: not beyond one side; 1: beyond the low side; 2: beyond the high side
        if (lo >= (int)0xC0000000 && hi <= (int)0xFF800000) return 1;
        if (lo >= beyond_bits && hi < (int)f2u(2147483648.0f)) return 2;
        return 0;
      };
      // (the extremes are wave-uniform values in vector registers; what is derived from them and kept is made scalar)
      const int sx_side = __builtin_amdgcn_readfirstlane(side(w_lo_x, w_hi_x, beyond_x_bits));
      const int sy_side = __builtin_amdgcn_readfirstlane(side(w_lo_y, w_hi_y, beyond_y_bits));
      if (LRP_WIN_CORNER != 0 && sx_side != 0 && sy_side != 0) b.tier = (1 + (sx_side - 1) + 2 * (sy_side - 1)) << 3;
      // ... beyond one SIDE only (a rectilinear view inside a panorama: the rows above and below the view and the
      // columns left and right of it, another 37 % of the blocks): the same reasoning along one axis — all four tap
      // rows (columns) are the first or the last source row (column), the weight of that axis is 0 or 1 — and the
      // unclamped case along the other.  The block then reads ONE source row or column.
      else if (kEdge && P.win_edge != 0 && sy_side != 0 && in_x) {
        const int x_first = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_lo_x));
        const int x_last = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_hi_x));
        b.x_lo = x_first - 1;
        b.bw = x_last + 2 - b.x_lo + 1;
        b.y_lo = sy_side == 2 ? P.in_h - 1 : 0;
        b.bh = 1;
        b.pitch = b.bw;
        b.c_base = raw_slots(b); // the plane of vertical cubics, one per window texel (RGBAZ: + a float plane behind it)
        if (b.c_base + raw_slots(b) <= kCap) b.tier = 1 | ((1 + (sy_side - 1)) << 6);
      } else if (kEdge && P.win_edge != 0 && sx_side != 0 && in_y) {
        const int y_first = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_lo_y));
        const int y_last = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_hi_y));
        b.y_lo = y_first - 1;
        b.bh = y_last + 2 - b.y_lo + 1;
        b.x_lo = sx_side == 2 ? P.in_w - 1 : 0;
        b.bw = 1;
        b.pitch = 1;
        if (raw_slots(b) <= kCap) b.tier = 1 | ((3 + (sx_side - 1)) << 6);
      }
    }
  };
  auto clear_block = [](WinBlock &b) {
    b.tier = 0;
    b.x_lo = b.y_lo = b.bw = b.bh = b.pitch = b.c_plane = b.c_base = b.tap_base = 0;
    b.y_lo2 = b.bh2 = 0;
    b.c_delta_stored[0] = b.c_delta_stored[1] = 0;
    b.iy0[0] = b.iy0[1] = b.iyn[0] = b.iyn[1] = 0;
  };
  // One pixel's contribution to the extremes.  Per pixel only the exactness half of interior()
  // (it also fails for NaN / inf); the range half is voted on the wave-wide extremes.  For finite
  // floats the raw bits order like signed integers as long as the minimum is >= 0, and a negative
  // coordinate makes the signed minimum negative, so v_min_i32 / v_max_i32 on the bits give the
  // extremes (no canonicalising float min / max); a NaN is a huge or a negative integer and fails
  // the range vote as well.
  struct Extremes {
    int lo_x = 0x7fffffff, hi_x = (int)0x80000000;
    int lo_y[2] = {0x7fffffff, 0x7fffffff}, hi_y[2] = {(int)0x80000000, (int)0x80000000};
    int exact_x = 1, exact_y = 1;
  };
  auto note_pixel = [](Extremes &e, int k, float sx, float sy) {
    const f2 sxy{sx, sy};
    const f2 back = (sxy + 2.0f) - sxy; // both coordinates in one packed add / subtract
    e.exact_x &= (int)(back.x == 2.0f);
    e.exact_y &= (int)(back.y == 2.0f);
    const int bx = (int)f2u(sx), by = (int)f2u(sy);
    e.lo_x = min(e.lo_x, bx);
    e.hi_x = max(e.hi_x, bx);
    e.lo_y[k >> 1] = min(e.lo_y[k >> 1], by);
    e.hi_y[k >> 1] = max(e.hi_y[k >> 1], by);
  };

  // Mirrored strips: the windows of all four mirror blocks from ONE pair of wave-wide reductions.
  // The source x of a pixel only depends on whether the block is mirrored in x, its source y on
  // whether it is mirrored in y, so the strip has two x ranges and (two halves x) two y ranges —
  // 12 extremes, two wave_box calls — instead of 4 blocks x 6.  They are parked in the lanes of one
  // VGPR (`plan`, lane i = extreme i as float bits) and fetched with v_readlane when a block starts:
  //   0-3:  x lo / hi unmirrored, x lo / hi mirrored
  //   4-11: y lo / hi of half a, of half b — unmirrored, then mirrored
  // plan_exact: bit 0 / 1 = every pixel's x + 2 exact (unmirrored / mirrored), bit 2 / 3 likewise for y.
  // A block whose cheap exactness vote failed (next to a power-of-two coordinate) is planned the
  // long way, precise test included.
  int plan = 0;
  uint32_t plan_exact = 0;
  constexpr bool kStripPlan = Quad && !kSharedRays && LRP_WIN_STRIP_PLAN != 0;
  auto plan_strip = [&]() {
    Extremes e0, e1; // unmirrored (image 0) and mirrored in every mirrored axis (an axis that is not mirrored has one range: e1's equals e0's)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float sx, sy;
      quad_xy(0, k, sx, sy);
      note_pixel(e0, k, sx, sy);
      quad_xy(kAllMirrors, k, sx, sy);
      note_pixel(e1, k, sx, sy);
    }
    plan_exact = (wave_all(e0.exact_x != 0) ? 1u : 0u) | (wave_all(e1.exact_x != 0) ? 2u : 0u) |
                 (wave_all(e0.exact_y != 0) ? 4u : 0u) | (wave_all(e1.exact_y != 0) ? 8u : 0u);
    wave_box(e0.lo_x, e0.hi_x, e1.lo_x, e1.hi_x, e0.lo_y[0], e0.hi_y[0]);
    wave_box(e0.lo_y[1], e0.hi_y[1], e1.lo_y[0], e1.hi_y[0], e1.lo_y[1], e1.hi_y[1]);
    const int v[12] = {e0.lo_x, e0.hi_x, e1.lo_x, e1.hi_x, e0.lo_y[0], e0.hi_y[0],
                       e0.lo_y[1], e0.hi_y[1], e1.lo_y[0], e1.hi_y[0], e1.lo_y[1], e1.hi_y[1]};
#pragma unroll
    for (int i = 0; i < 12; ++i) // (this clang has no writelane builtin)
      asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(plan) : "s"(v[i]), "n"(i));
  };

  // phase A of block g: coordinates, interior vote, window box
  auto coords = [&](int g, WinBlock &b) {
    // the four row terms first, all loads in flight together (one exposed latency per
    // block instead of one in front of every pixel's coordinate chain)
    const int gm = image_of(g); // (plain blocks: g)
    const int mx = quad ? (gm & 1) : 0, my = quad ? (gm >> 1) : 0;
    (void)mx;
    (void)my;
    float row_v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!quad || g == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yk = y_lane + (quad ? 0 : kBlkH * block_row(g)) + kPassRows * k;
        row_v[k] = row_term<OutLens>(P, yk < qh ? yk : qh - 1, 0);
      }
    }
    if (quad && g == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yk = y_lane + kPassRows * k;
        const int ye = yk < qh ? yk : qh - 1;
        float u, v;
        if constexpr (kSharedRays) {
          pixel_ray<OutLens>(P, col, row_v[k], ye, 0, qa[k], qb[k], qc[k]);
          (void)u;
          (void)v;
        } else if constexpr (kEqrByTheta) {
          // vec_to_equirectangular (src/reproject.cpp:259-271) split at the longitude: everything up to theta once, the
          // rest of :268 once per sign of theta; the latitude half is the same for both mirror images
          float vx, vy, vz;
          pixel_ray<OutLens>(P, col, row_v[k], ye, 0, vx, vy, vz);
          if (P.has_rot) { // :303-311
            const float nx = P.rot[0] * vx + P.rot[1] * vy + P.rot[2] * vz;
            const float ny = P.rot[3] * vx + P.rot[4] * vy + P.rot[5] * vz;
            const float nz = P.rot[6] * vx + P.rot[7] * vy + P.rot[8] * vz;
            vx = nx;
            vy = ny;
            vz = nz;
          }
          const float lon_min = P.in_lens.p[2], img_w = (float)P.in_w;
          const float theta = -atan2f_(-vx, -vz); // :262
          qa[k] = texel_coord(((theta - lon_min) / P.in_lon_span - 0.5f) * img_w, img_w);  // :268, :323
          qc[k] = texel_coord(((-theta - lon_min) / P.in_lon_span - 0.5f) * img_w, img_w); // the mirrored pixel's
          // the centre column of an odd-sized image is its own mirror image: nx is a zero, theta is 0 or +-pi, and -pi is
          // not the same longitude bit for bit — the pixel is rendered twice, both times with its own theta
          if (2 * xe == P.out_w - 1) qc[k] = qa[k];
          qb[k] = texel_coord(equirect_cy(vx, vy, vz, P.in_lens.p[0], P.in_lat_span, (float)P.in_h), (float)P.in_h); // :263, :269, :324
          (void)u;
          (void)v;
          // one pixel after the other: interleaved, the four atan2f / asinf evaluations need more registers than there are
          __builtin_amdgcn_sched_barrier(0);
        } else {
          pixel_plane<OutLens, InMode>(P, col, row_v[k], ye, 0, u, v);
          if constexpr (kInEqr) { // host guarantees the xsep table: v = phi
            float unused;
            plane_to_texel<OutLens, InMode>(P, col, u, v, unused, qa[k]);
            plane_to_texel<OutLens, InMode>(P, col, u, -v, unused, qb[k]);
          } else {
            qa[k] = u;
            qb[k] = v;
          }
        }
      }
      if constexpr (kStripPlan) plan_strip();
    }
    clear_block(b);
    if constexpr (kStripPlan) {
      if (((plan_exact >> mx) & (plan_exact >> (2 + my)) & 1u) != 0) {
        plan_window(b, __builtin_amdgcn_readlane(plan, 2 * mx), __builtin_amdgcn_readlane(plan, 2 * mx + 1),
                    __builtin_amdgcn_readlane(plan, 4 + 4 * my), __builtin_amdgcn_readlane(plan, 5 + 4 * my),
                    __builtin_amdgcn_readlane(plan, 6 + 4 * my), __builtin_amdgcn_readlane(plan, 7 + 4 * my), true, true);
        return;
      }
    }
    Extremes e;
    // (plain blocks: everything derived from the column terms alone — their products with the rotation matrix, the
    // column's share of the source lens — is loop-invariant, gets hoisted out of the block loop and then spilled to scratch
    // for the whole kernel: 80-100 MB of scratch traffic per 4K frame.  Opaque here, those few multiplies run per block.)
    ColTerms col_g = col;
    if constexpr (!Quad && LRP_OPAQUE_COL != 0) asm volatile("" : "+v"(col_g.a), "+v"(col_g.b), "+v"(col_g.nx), "+v"(col_g.nz), "+v"(col_g.sx));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yk = y_lane + (quad ? 0 : kBlkH * block_row(g)) + kPassRows * k;
      const int ye = yk < qh ? yk : qh - 1;
      if (!quad)
        pixel_source_rt<OutLens, InMode>(P, col_g, row_v[k], ye, 0, b.sx[k], b.sy[k]);
      else
        quad_xy(gm, k, b.sx[k], b.sy[k]);
      note_pixel(e, k, b.sx[k], b.sy[k]);
      // (shared rays: one pixel's rotation + source lens after the other — interleaved they do not fit the registers)
      if constexpr (kSharedRays) __builtin_amdgcn_sched_barrier(0);
    }
    bool all_exact_x, all_exact_y;
    if constexpr (kEdge) { // per axis: a block beyond one side of the source has no exact taps along that axis and needs none
      all_exact_x = wave_all(e.exact_x != 0);
      all_exact_y = wave_all(e.exact_y != 0);
      if (!(all_exact_x && all_exact_y)) { // the precise test, for the stripe of blocks next to a power-of-two coordinate
        int ok_x = 1, ok_y = 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          ok_x &= taps_consecutive(b.sx[k], 2.0f);
          ok_y &= taps_consecutive(b.sy[k], 2.0f);
        }
        all_exact_x = wave_all(ok_x != 0);
        all_exact_y = wave_all(ok_y != 0);
      }
    } else {
      bool all_exact = wave_all((e.exact_x & e.exact_y) != 0);
      if (!all_exact) { // the precise test, for the stripe of blocks next to a power-of-two coordinate
        int ok = 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) ok &= taps_consecutive(b.sx[k], 2.0f) & taps_consecutive(b.sy[k], 2.0f);
        all_exact = wave_all(ok != 0);
      }
      all_exact_x = all_exact_y = all_exact;
    }
    const bool planned = kEdge ? true : (all_exact_x && all_exact_y);
    if (planned) {
      wave_box(e.lo_x, e.hi_x, e.lo_y[0], e.hi_y[0], e.lo_y[1], e.hi_y[1]);
      plan_window(b, e.lo_x, e.hi_x, e.lo_y[0], e.hi_y[0], e.lo_y[1], e.hi_y[1], all_exact_x, all_exact_y);
    }
    if constexpr (kGeoWrite) {
      if (geo_write) { // side output: this block's coordinates and the extremes its window was planned from
        vf2 *const map = reinterpret_cast<vf2 *>(P.geo_xy);
        if (Pk.geo_mode == 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int yk = y_lane + kBlkH * block_row(g) + kPassRows * k;
            const int ye = yk < qh ? yk : qh - 1;
            // (lanes / rows beyond the image hold the pixel they were clamped to and write its values to its place again)
            map[geo_map_index(xe, ye, P.out_w)] = vf2{b.sx[k], b.sy[k]};
          }
        }
        const int words[7] = {planned ? e.lo_x : 0, planned ? e.hi_x : 0, planned ? e.lo_y[0] : 0, planned ? e.hi_y[0] : 0,
                              planned ? e.lo_y[1] : 0, planned ? e.hi_y[1] : 0,
                              (all_exact_x ? 1 : 0) | (all_exact_y ? 2 : 0) | (planned ? 4 : 0)};
        int bv = 0; // lane i = word i (written once per geometry: plain selects will do)
#pragma unroll
        for (int i = 0; i < 7; ++i) bv = lane == i ? words[i] : bv;
        if (lane < 8) P.geo_box[geo_block(g) * 8u + (uint32_t)lane] = bv;
      }
    }
  };
  // GeoRead: the coordinates of block g and (geo_boxv, lane i = word i) its window extremes are requested by geo_fetch and
  // turned into a window plan by geo_plan, a few hundred instructions later
  int geo_boxv = 0;
  auto geo_fetch = [&](int g, WinBlock &b) {
    // (the extremes first: loads return in order, and the first block of a strip plans its window before anything else)
    geo_boxv = __builtin_nontemporal_load(P.geo_box + (geo_block(g) * 8u + (uint32_t)(lane & 7)));
    const vf2 *const map = reinterpret_cast<const vf2 *>(P.geo_xy);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yk = y_lane + kBlkH * block_row(g) + kPassRows * k;
      const int ye = yk < qh ? yk : qh - 1;
      const vf2 v = __builtin_nontemporal_load(map + geo_map_index(xe, ye, P.out_w));
      b.sx[k] = v.x;
      b.sy[k] = v.y;
    }
  };
  auto geo_plan = [&](WinBlock &b) {
    clear_block(b);
    const int flags = __builtin_amdgcn_readlane(geo_boxv, 6);
    if ((flags & 4) != 0)
      plan_window(b, __builtin_amdgcn_readlane(geo_boxv, 0), __builtin_amdgcn_readlane(geo_boxv, 1),
                  __builtin_amdgcn_readlane(geo_boxv, 2), __builtin_amdgcn_readlane(geo_boxv, 3),
                  __builtin_amdgcn_readlane(geo_boxv, 4), __builtin_amdgcn_readlane(geo_boxv, 5), (flags & 1) != 0, (flags & 2) != 0);
  };
  // the one value of a corner block: sample_bicubic with all 16 taps on the corner texel (sample_direct's
  // one-column-and-one-row case, same operations)
  auto corner_value = [&](const WinBlock &b) {
    const int xh = (b.corner() - 1) & 1, yh = (b.corner() - 1) >> 1;
    const float fx = xh ? 1.0f : 0.0f, fy = yh ? 1.0f : 0.0f; // the clamped weights (src/reproject.cpp:130-131)
    const float hfx = 0.5f * fx, hfy = 0.5f * fy;
    const uint32_t off = (uint32_t)(yh ? P.in_h - 1 : 0) * src.row_bytes + (uint32_t)(xh ? in_w - 1 : 0) * (4u * CH);
    if constexpr (CH == 5) {
      const Px<5> t = texel_at<5>(src.rsrc, off, 0u);
      const Px<5> k = cubic_px<5>(t, t, t, t, fy, hfy);
      const Px<5> r = cubic_px<5>(k, k, k, k, fx, hfx);
      return Rgba{r.lo, r.hi, r.e};
    } else {
      const Px<4> t = texel_at<4>(src.rsrc, off, 0u); // (RGB: a 16-byte read of a 12-byte texel, fourth component unused)
      const Px<4> k = cubic_px<4>(t, t, t, t, fy, hfy);
      return cubic_px<4>(k, k, k, k, fx, hfx);
    }
  };
  auto issue = [&](const float *frame, const WinBlock &b, int half = 0) { // the window `b` of the source frame `frame` (split blocks: of its half)
    if (kEdge && b.edge() != 0) {
      // One source row (texels x_lo .. x_lo + bw - 1 of row y_lo) or one source column (rows y_lo .. y_lo + bh - 1 of
      // column x_lo) into consecutive slots: 64 texels per instruction, the lane's byte offset along the row / column
      // in a VGPR, the first texel's address in an SGPR pair.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const bool along_y = b.edge() >= 3;
      const int n = along_y ? b.bh : b.bw;
      const uint32_t step = along_y ? src.row_bytes : 4u * CH;
      const uint32_t first_row = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)b.y_lo * src.row_bytes));
      const uint32_t first_col = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)b.x_lo * (4u * CH)));
      const char *first = reinterpret_cast<const char *>(frame) + ((size_t)first_row + (size_t)first_col);
      const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)win0;
      for (int c0 = 0; c0 < n; c0 += 64) {
        if (c0 + lane < n) {
          const uint32_t lane_bytes = (uint32_t)(c0 + lane) * step;
          const uint32_t lds = lds0 + (uint32_t)c0 * 16u;
          if constexpr (CH == 3)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %2"
                         :
                         : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(lane_bytes), "s"(first)
                         : "memory", "m0");
          else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :
                         : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(lane_bytes), "s"(first)
                         : "memory", "m0");
          if constexpr (CH == 5) { // depth: the float plane behind the n colour slots
            const uint32_t lds_d = lds0 + (uint32_t)n * 16u + (uint32_t)c0 * 4u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
                         :
                         : "s"(__builtin_amdgcn_readfirstlane(lds_d)), "v"(lane_bytes), "s"(first + 16)
                         : "memory", "m0");
          }
        }
      }
    } else if (b.staged()) {
      // LDS-DMA, one window row per instruction, lanes beyond the width masked off.  The address is a wave-uniform row
      // base in an SGPR pair (advanced by scalar adds) plus one per-lane byte offset that is the same for every row and
      // every frame: no vector arithmetic per row.
      float4 *const win = win0;
      // Issued as inline assembly: the compiler's wait-count insertion then does not know
      // that LDS is being written and puts no vmcnt(0) in front of later LDS reads (of the
      // coefficient planes, which the DMA does not touch); the one wait that IS needed sits
      // at the top of the block loop.  M0 = LDS byte address of the row (+ 16 B per lane).
      // the reads of the window issued so far have returned before anything overwrites it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // (windows wider than 64 texels — pass windows of the big-window variant — take one instruction per 64 columns and row)
      const int n_chunks = kMaxPassCols > 64 ? (b.bw + 63) >> 6 : 1;
      for (int chunk = 0; chunk < n_chunks; ++chunk)
      if (chunk * 64 + lane < b.bw) {
        uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)win + (uint32_t)(b.org() + chunk * 64) * 16u;
        const uint32_t lds_step = (uint32_t)(b.spitch() * 16); // dwordx3 too writes one 16-byte slot per lane
        const uint32_t lane_bytes = (uint32_t)(b.x_lo + chunk * 64 + lane) * (4u * CH);
        const int n_rows = kSplit ? b.rows_of(half) : b.bh;
        const char *row = reinterpret_cast<const char *>(frame) + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(kSplit ? b.first_row_of(half) : b.y_lo) * src.row_bytes)); // wave-uniform
        for (int r = 0; r < n_rows; ++r) {
          if constexpr (CH == 4)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :
                         : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(lane_bytes), "s"(row)
                         : "memory", "m0");
          else if constexpr (CH == 5) {
            // colour into the 16-byte slots of the row, depth into the row of the float plane behind the colour plane
            // (an instruction offset would move the LDS address as well as the global one: the fifth float's 16 bytes
            // go into the scalar base)
            const uint32_t lds_d = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)win +
                                   (uint32_t)(b.pitch * n_rows) * 16u + (uint32_t)(r * b.pitch + chunk * 64) * 4u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :
                         : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(lane_bytes), "s"(row)
                         : "memory", "m0");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
                         :
                         : "s"(__builtin_amdgcn_readfirstlane(lds_d)), "v"(lane_bytes), "s"(row + 16)
                         : "memory", "m0");
          } else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %2"
                         :
                         : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(lane_bytes), "s"(row)
                         : "memory", "m0");
          lds += lds_step;
          row += src.row_bytes;
        }
      }
    }
  };

  // Coefficient tier.  Of the 17 operations of a vertical Catmull-Rom evaluation, 11
  // depend on the four taps only, not on the weight:
  //     inner = ((3 (b - c)) + d) - a,  m0 = (((2 a - 5 b) + 4 c) - d),  cma = c - a
  //     k = b + hfy * (cma + fy * (m0 + fy * inner))          (src/reproject.cpp:92-98)
  // Under magnification the pixels of a block share their tap columns, so those three
  // terms are evaluated ONCE per tap-column origin of the window (a lane per origin,
  // straight from the staged window) and stored behind it in three planes; a pixel
  // then spends 6 instead of 17 operations per channel and column.  Same operations
  // on the same operands in the same order: the result is the reference's, bit for bit.
  auto precompute = [&](const WinBlock &b, int h) {
    const int n = b.pitch * b.iyn[h]; // origins: every window column x every first tap row of this half
    // Origins are enumerated over their contiguous slot range from its lowest slot (bottom-up storage
    // puts the LAST origin row there); an origin's four taps are spitch slots apart either way.
    const float4 *const raw = win0 + (b.org() + (b.iy0[h] - 1 - b.y_lo) * b.spitch() + (b.spitch() < 0 ? (b.iyn[h] - 1) * b.spitch() : 0));
    float4 *const planes = win0 + b.c_base;
    auto load4 = [&](int idx, Rgba t[4]) { // origin idx = row * pitch + column reads the window texels idx + {0, 1, 2, 3} * pitch
      const float4 *q = raw + (idx < n ? idx : n - 1); // (origins in the pad column of an odd pitch compute unused values from stale slots)
      t[0] = as_rgba(q[0]);
      t[1] = as_rgba(q[b.spitch()]);
      t[2] = as_rgba(q[2 * b.spitch()]);
      t[3] = as_rgba(q[3 * b.spitch()]);
    };
    auto emit = [&](int idx, const Rgba t[4]) {
      Rgba inner, m0, cma;
      inner.lo = ((3.0f * (t[1].lo - t[2].lo)) + t[3].lo) - t[0].lo;
      inner.hi = ((3.0f * (t[1].hi - t[2].hi)) + t[3].hi) - t[0].hi;
      m0.lo = (((2.0f * t[0].lo) - (5.0f * t[1].lo)) + (4.0f * t[2].lo)) - t[3].lo;
      m0.hi = (((2.0f * t[0].hi) - (5.0f * t[1].hi)) + (4.0f * t[2].hi)) - t[3].hi;
      cma.lo = t[2].lo - t[0].lo;
      cma.hi = t[2].hi - t[0].hi;
      if (idx < n) {
        planes[idx] = float4{inner.lo.x, inner.lo.y, inner.hi.x, inner.hi.y};
        planes[b.c_plane + idx] = float4{m0.lo.x, m0.lo.y, m0.hi.x, m0.hi.y};
        planes[2 * b.c_plane + idx] = float4{cma.lo.x, cma.lo.y, cma.hi.x, cma.hi.y};
      }
    };
    // two chunks of 64 origins per trip, the second chunk's reads in flight under the first chunk's arithmetic
#pragma unroll 1
    for (int i0 = 0; i0 < n; i0 += 128) {
      const bool two = i0 + 64 < n; // wave-uniform
      Rgba t0[4], t1[4];
      load4(i0 + lane, t0);
      if (two) load4(i0 + 64 + lane, t1);
      emit(i0 + lane, t0);
      if (two) emit(i0 + 64 + lane, t1);
    }
  };

  // Blocks beyond the first / last source row: k = cubic(t, t, t, t, fy) of every window texel t, fy = 0 / 1 the clamped
  // weight (src/reproject.cpp:131), a lane per texel, into the plane behind the row.  Evaluated, not assumed to be t:
  // with a non-finite texel it is not.
  auto edge_plane = [&](const WinBlock &b) {
    const float fyc = b.edge() == 2 ? 1.0f : 0.0f, hfyc = 0.5f * fyc;
    float4 *const plane = win0 + b.c_base;
    const float *const raw_d = reinterpret_cast<const float *>(win0 + b.bw);
    float *const plane_d = reinterpret_cast<float *>(plane + b.bw);
#pragma unroll 1
    for (int i0 = 0; i0 < b.bw; i0 += 64) {
      const int i = min(i0 + lane, b.bw - 1);
      const Rgba t = as_rgba(win0[i]);
      const Rgba k = cubic4(t, t, t, t, fyc, hfyc);
      plane[i] = float4{k.lo.x, k.lo.y, k.hi.x, k.hi.y};
      if constexpr (CH == 5) {
        const float dz = raw_d[i];
        plane_d[i] = catmull_rom(dz, dz, dz, dz, fyc, hfyc);
      }
    }
  };

  // vmcnt retires in order, stores included: a store issued BEFORE the DMA of the next window
  // would have to be acknowledged by memory before that window counts as landed.  So the
  // DMA of window g+1 is issued inside the last pass of block g, right behind that pass's
  // reads of the window and ahead of its arithmetic and its store; the stores of passes
  // 0-2 are a pass or more old by then, the store of pass 3 is the one vm operation that
  // may still be outstanding when the next block waits: vmcnt(1).  (Every lane stores,
  // see below, so that store is always issued.)
  WinBlock cur, nxt;
  if constexpr (GeoRead) {
    geo_fetch(0, cur);
    geo_plan(cur);
  } else {
    coords(0, cur);
  }
  issue(P.src, cur);
  int g_loop = 0, f_loop = 0;
  bool dma_early = false; // the pending window was requested before its block's last store
  // The step after (block g_loop, frame f_loop): the same block in the next frame, or the next block in the first frame.
  auto issue_next = [&]() {
    if (f_loop + 1 < n_frames)
      issue(frame_src(f_loop + 1), cur);
    else
      issue(frame_src(0), nxt);
  };
  auto has_next = [&]() { return f_loop + 1 < n_frames || g_loop + 1 < G; };
  auto next_window = [&]() {
    // while this block's coefficient planes are still being read the next raw window must stay in front of them
    // (the same block's window in the next frame always does: planes sit behind the raw window)
    // (likewise the plane of vertical cubics of a block beyond the first / last source row: edge() 1, 2)
    const bool planes_live = (kWinCoef && cur.coef()) || (kEdge && cur.edge() != 0 && cur.edge() < 3);
    dma_early = has_next() && (!planes_live || f_loop + 1 < n_frames || raw_slots(nxt) <= cur.c_base);
    if (dma_early) issue_next();
  };
  // The result of pass k of block g: num_samples == 1, (0.0f + s) * normalize (src/reproject.cpp:334-341), store.
  auto emit = [&](int g, int k, const Rgba &s, auto as_runs, bool runs_rt = true) {
    Rgba a4 = px_zero<4>();
    px_add<4>(a4, s);
    if constexpr (CH == 5) a4.e = 0.0f + s.e;
    const Px<CH> a{a4.lo, CH >= 4 ? a4.hi : f2{0.0f, 0.0f}, CH == 3 ? a4.hi.x : a4.e};
    // Every lane stores: lanes / rows beyond the image have recomputed the pixel they were
    // clamped to (xe, ye) and write that same value to that same address again, so the
    // store is issued by every wavefront (the vmcnt(1) below counts on it).
    // (the four clamped rows of a mirrored strip are loop-invariant; hoisted they occupy four VGPRs for the whole
    // strip — which spilled — so the row is re-derived from an opaque copy here: an add and a min per pass)
    int y_base = y_lane;
    asm volatile("" : "+v"(y_base)); // (likewise not hoisted out of the frame loop)
    const int yk = y_base + (quad ? 0 : kBlkH * block_row(g)) + kPassRows * k;
    const int yc = yk < qh ? yk : qh - 1;
    const int gm = image_of(g);
    const int xo = (quad && (gm & 1)) ? P.out_w - 1 - xe : xe; // mirrored blocks write the mirrored pixel
    const int yo = (quad && (gm >> 1)) ? P.out_h - 1 - yc : yc;
#if defined(LRP_NO_STORE) // timing experiment: almost no output traffic
    if (a.lo.x == 12345.678f) store_px<CH, true>(P, (uint32_t)yo * (uint32_t)P.out_w + (uint32_t)xo, a);
#else
    if constexpr (CH == 5) {
     if constexpr (decltype(as_runs)::value) {
      // a pass that lies in the image whole (wave-uniform) leaves as four runs of 16 pixels (store_rgbaz_run).
      // Used where the stores are what a block costs: corner blocks (four stores and nothing else) in every kernel,
      // all blocks of a rectilinear view rendered into a panorama (kRunsEverywhere: most of that frame is out of
      // view or gathers minified taps; 383 -> 334 us).  In the VALU-bound kernels that interpolate from the LDS
      // window the exchange costs more than the stores gain (measured: 4-7 % slower).
      const int x_blk = tx * (kBlkW * kWinWaves) + wave * kBlkW;
      const int y_top = P.y_offset + ty * (quad ? kBlkH : kBlkH * Gs) + (quad ? 0 : kBlkH * block_row(g)) + kPassRows * k;
      if (runs_rt && x_blk + kBlkW <= qw && y_top + kPassRows <= qh) {
        const bool mxo = quad && (gm & 1), myo = quad && (gm >> 1);
        float c[5];
        finish_px<5, true>(P, a, c);
        const uint32_t first = (uint32_t)(myo ? P.out_h - 1 - y_top : y_top) * (uint32_t)P.out_w +
                               (uint32_t)(mxo ? P.out_w - x_blk - kBlkW : x_blk);
        store_rgbaz_run<4>(P, out_lds, prow * kBlkW + (mxo ? kBlkW - 1 - pcol : pcol), first, myo ? -P.out_w : P.out_w, c);
        return;
      }
     }
    }
    store_px<CH, true>(P, (uint32_t)yo * (uint32_t)P.out_w + (uint32_t)xo, a);
#endif
  };
  // RGBAZ: the depth channel of one pixel from the float plane behind the colour window.  `slot` is the
  // window slot of the pixel's first tap (int(sx) - 1, int(sy) - 1); bicubicInterpolate's order: four
  // vertical cubics, then the horizontal one (src/reproject.cpp:100-107).  In the last pass these are the
  // block's last reads of the window: the next window's DMA goes behind them.
  auto depth_from_window = [&](const float4 *win, int slot, float fx, float fy, float hfx, float hfy, bool last_pass, int half = 0) {
    const float *d = reinterpret_cast<const float *>(win + cur.pitch * (kSplit ? cur.rows_of(half) : cur.bh)) + slot;
    float t[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j][r] = d[r * cur.pitch + j];
    if (last_pass) next_window();
    // the four vertical cubics as two packed ones (tap columns 0 | 1 and 2 | 3 in the halves of a register pair:
    // each half of a v_pk_* rounds like the scalar instruction), then the horizontal one
    const f2 k01 = catmull_rom2(f2{t[0][0], t[1][0]}, f2{t[0][1], t[1][1]}, f2{t[0][2], t[1][2]}, f2{t[0][3], t[1][3]}, fy, hfy);
    const f2 k23 = catmull_rom2(f2{t[2][0], t[3][0]}, f2{t[2][1], t[3][1]}, f2{t[2][2], t[3][2]}, f2{t[2][3], t[3][3]}, fy, hfy);
    return catmull_rom(k01.x, k01.y, k23.x, k23.y, fx, hfx);
  };
  // Pass windows (kernels with split blocks): a block whose two half windows do not fit either (a pole face of a
  // cubemap: the panorama's rows converge) still has passes — 16 x 4 pixels — whose own window fits.  Planned per pass
  // from the wave-wide extremes of that pass's coordinates, fetched, waited for and read on the spot; the other
  // wavefronts of the SIMD cover the round trip (requesting the window of pass k + 1 behind the taps of pass k, like the
  // second half of a split block, measured slower: 120 against 112 us per pole face, 14 spilled registers).
  // False: this pass gathers per pixel.
  // taps of one pixel from a staged pass window `w` and the five cubics (bicubicInterpolate's order, src/reproject.cpp:100-107)
  auto window_sample = [&](const WinBlock &w, float psx, float psy, bool last_pass) -> Rgba {
    const float4 *const win = win0;
    const float tx_ = __builtin_truncf(psx), ty_ = __builtin_truncf(psy);
    const float fx = psx - tx_, fy = psy - ty_;
    const int slot0 = __mul24((int)ty_ - 1 - w.y_lo, w.pitch) + ((int)tx_ - 1 - w.x_lo);
    const float4 *t = win + slot0;
    const float hfx = 0.5f * fx, hfy = 0.5f * fy;
    const float4 *t1 = t + w.pitch, *t2 = t1 + w.pitch, *t3 = t2 + w.pitch;
    Rgba q[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      q[j][0] = as_rgba(t[j]);
      q[j][1] = as_rgba(t1[j]);
      q[j][2] = as_rgba(t2[j]);
      q[j][3] = as_rgba(t3[j]);
    }
    float dz[4][4];
    if constexpr (CH == 5) {
      const float *d = reinterpret_cast<const float *>(win + w.pitch * w.bh) + slot0;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) dz[j][r] = d[r * w.pitch + j];
    }
    if (last_pass) next_window(); // behind the last reads of this pass's window
    const Rgba k0 = cubic4(q[0][0], q[0][1], q[0][2], q[0][3], fy, hfy);
    const Rgba k1 = cubic4(q[1][0], q[1][1], q[1][2], q[1][3], fy, hfy);
    const Rgba k2 = cubic4(q[2][0], q[2][1], q[2][2], q[2][3], fy, hfy);
    const Rgba k3 = cubic4(q[3][0], q[3][1], q[3][2], q[3][3], fy, hfy);
    Rgba s = cubic4(k0, k1, k2, k3, fx, hfx);
    if constexpr (CH == 5) {
      const f2 k01 = catmull_rom2(f2{dz[0][0], dz[1][0]}, f2{dz[0][1], dz[1][1]}, f2{dz[0][2], dz[1][2]}, f2{dz[0][3], dz[1][3]}, fy, hfy);
      const f2 k23 = catmull_rom2(f2{dz[2][0], dz[3][0]}, f2{dz[2][1], dz[3][1]}, f2{dz[2][2], dz[3][2]}, f2{dz[2][3], dz[3][3]}, fy, hfy);
      s.e = catmull_rom(k01.x, k01.y, k23.x, k23.y, fx, hfx);
    }
    return s;
  };
  // window of one pass (or of two passes that read the same source rows) from the wave-wide extremes of its coordinates;
  // false: too wide or too large for the buffer
  auto plan_pass_window = [&](WinBlock &w, int lo_x, int hi_x, int lo_y, int hi_y) -> bool {
    int d0 = 0, d1 = 0;
    wave_box(lo_x, hi_x, lo_y, hi_y, d0, d1); // (interior: the coordinates are >= 1, their bits order like integers)
    clear_block(w);
    w.x_lo = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)lo_x)) - 1;
    w.y_lo = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)lo_y)) - 1;
    w.bw = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)hi_x)) + 2 - w.x_lo + 1;
    w.bh = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)hi_y)) + 2 - w.y_lo + 1;
    w.pitch = w.bw | 1;
    if (w.bw > kMaxPassCols || slots_of_rows(w.pitch, w.bh) > kCap) return false;
    w.tier = 1;
    return true;
  };
  auto pass_window = [&](float psx, float psy, Rgba &s, bool last_pass) -> bool {
    if (!all_interior(psx, psy, 1.0f, src.x_hi, src.y_hi, 2.0f)) return false;
    WinBlock w;
    if (!plan_pass_window(w, (int)f2u(psx), (int)f2u(psx), (int)f2u(psy), (int)f2u(psy))) return false;
    issue(P.src, w);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the window (and every older store)
    s = window_sample(w, psx, psy, last_pass);
    return true;
  };
#pragma unroll 1
  for (int g = 0; g < G; ++g) {
   g_loop = g;
   // plain blocks: the next block's coordinates here, long before its window is requested in the
   // last pass; mirrored blocks derive theirs in a few instructions right there (fewer live registers)
   // (GeoRead: the next block's record is requested behind this block's wait and planned in front of its last pass — the
   // loads are then older than the next window's DMA and the hand-counted vmcnt(1) below still holds)
   if constexpr (!GeoRead)
     if ((!Quad || kSharedRays) && g + 1 < G) coords(g + 1, nxt); // (shared rays: the rotation and the source lens run per image, as for a plain block)
#pragma unroll 1
   for (int f = 0; f < n_frames; ++f) {
    f_loop = f;
    if (n_frames > 1 || g == 0) set_frame(f);
    const bool last_frame = f + 1 == n_frames; // the next step is the next block
#if !defined(LRP_NO_DMA_WAIT) // timing experiment (wrong results): how much of the frame is exposed DMA / store latency
    // (a launch that writes the geometry cache has the stores of coords(g + 1) in flight as well: it waits for everything)
    if ((g == 0 && f == 0) || !dma_early || geo_write)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the window was the last thing requested
    else
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); // window g has landed; block g-1's last store may be in flight
#endif
    if constexpr (GeoRead)
      if (g + 1 < G && f == 0) geo_fetch(g + 1, nxt);
    const float4 *const win = win0;
    // The tier of this block in a scalar register for the branches below: carried through the block loop inside `cur` it
    // ends up in a VGPR (the kernel is at the SGPR limit), and every test of it then costs a v_and + v_cmp and the
    // branch condition is re-materialised through v_cndmask / v_cmp at each use — seven VALU instructions per pass.
#if LRP_OPT_TIER
    const int tier = __builtin_amdgcn_readfirstlane(cur.tier);
#else
    const int tier = cur.tier;
#endif
    const bool t_coef = kWinCoef && (tier & 2) != 0, t_staged = (tier & 1) != 0, t_whole = (tier & 4) != 0;
    const int t_edge = kEdge ? ((tier >> 6) & 7) : 0;
    const bool t_split = kSplit && (tier & 512) != 0; // the window holds passes 0-1; that of passes 2-3 is fetched behind pass 1's taps // 1, 2: beyond the first / last source row; 3, 4: column
#if defined(LRP_TIER_STATS)
    if (lane == 0) atomicAdd(&g_tier_stats[((tier >> 3) & 7) != 0 ? 3 : (tier & 512) != 0 ? 6 : ((tier >> 6) & 7) != 0 ? (((tier >> 6) & 7) < 3 ? 4 : 5) : (kWinCoef && (tier & 2)) ? 0 : (tier & 1) ? 1 : 2], 1u);
#endif
    if (((tier >> 3) & 7) != 0) {
      // every pixel of this block is one value: no taps, no per-pixel arithmetic — four stores.  The next block's
      // window is requested in front of the last store, as in the last pass of an ordinary block.
      const Rgba cs = corner_value(cur);
      if (Quad && !kSharedRays && last_frame && g + 1 < G) coords(g + 1, nxt);
      if constexpr (GeoRead)
        if (g + 1 < G && last_frame) geo_plan(nxt);
      emit(g, 0, cs, std::true_type{});
      emit(g, 1, cs, std::true_type{});
      emit(g, 2, cs, std::true_type{});
      next_window();
      emit(g, 3, cs, std::true_type{});
      if (!dma_early && has_next()) issue_next();
      if (last_frame) cur = nxt;
      continue;
    }
    if (t_edge == 1 || t_edge == 2) edge_plane(cur);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#if !defined(LRP_SKIP_PLANES) // timing experiment (wrong results): the coefficient phase removed
      if (t_coef && (h == 0 || !t_whole)) precompute(cur, h);
#endif
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int k = 2 * h + kk;
        if (Quad && !kSharedRays && k == 3 && last_frame && g + 1 < G) coords(g + 1, nxt); // only its box is kept
        if constexpr (GeoRead)
          if (k == 3 && g + 1 < G && last_frame) geo_plan(nxt);
        const bool last_pass = k == 3;
        float psx = cur.sx[k], psy = cur.sy[k];
        if constexpr (Quad && !kSharedRays) quad_xy(image_of(g), k, psx, psy); // re-derived (2-4 instructions) instead of held in registers
        // (where the coordinates of a mirror image are a plain selection of stored values the compiler would otherwise
        // hoist everything derived from them — truncations, weights, window addresses of all four passes and both
        // images — out of the block loop and spill it: the selected values are opaque here)
        // (the same goes for the frame loop: everything derived from the coordinates of a pass is the same in every frame,
        // and kept for all four passes it does not fit the registers — what IS shared between frames is stage 1 and the
        // window plan, by construction)
        asm volatile("" : "+v"(psx), "+v"(psy));
        Rgba s;
        if (t_edge != 0) {
          // sample_bicubic with one axis clamped (src/reproject.cpp:114-147): beyond the first / last source row the four
          // taps of a column are one texel t and its vertical cubic k = cubic(t, t, t, t, 0 or 1) comes from the plane,
          // leaving the horizontal cubic; beyond the first / last source column the four columns are one, their common
          // vertical cubic K is evaluated once and the horizontal one is cubic(K, K, K, K, 0 or 1).
          if (t_edge < 3) {
            const float tx_ = __builtin_truncf(psx);
            const float fx = psx - tx_, hfx = 0.5f * fx;
            const int slot = (int)tx_ - 1 - cur.x_lo;
            const float4 *kp = win + cur.c_base + slot;
            const Rgba k0 = as_rgba(kp[0]), k1 = as_rgba(kp[1]), k2 = as_rgba(kp[2]), k3 = as_rgba(kp[3]);
            float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (CH == 5) {
              const float *kd = reinterpret_cast<const float *>(win + cur.c_base + cur.bw) + slot;
#pragma unroll
              for (int j = 0; j < 4; ++j) d[j] = kd[j];
            }
            if (last_pass) next_window();
            s = cubic4(k0, k1, k2, k3, fx, hfx);
            if constexpr (CH == 5) s.e = catmull_rom(d[0], d[1], d[2], d[3], fx, hfx);
          } else {
            const float ty_ = __builtin_truncf(psy);
            const float fy = psy - ty_, hfy = 0.5f * fy;
            const float fxc = t_edge == 4 ? 1.0f : 0.0f, hfxc = 0.5f * fxc; // the clamped weight (:130)
            const int slot = (int)ty_ - 1 - cur.y_lo;
            const float4 *t = win + slot;
            const Rgba t0 = as_rgba(t[0]), t1 = as_rgba(t[1]), t2 = as_rgba(t[2]), t3 = as_rgba(t[3]);
            float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (CH == 5) {
              const float *td = reinterpret_cast<const float *>(win + cur.bh) + slot;
#pragma unroll
              for (int j = 0; j < 4; ++j) d[j] = td[j];
            }
            if (last_pass) next_window();
            const Rgba kk = cubic4(t0, t1, t2, t3, fy, hfy);
            s = cubic4(kk, kk, kk, kk, fxc, hfxc);
            if constexpr (CH == 5) {
              const float kd = catmull_rom(d[0], d[1], d[2], d[3], fy, hfy);
              s.e = catmull_rom(kd, kd, kd, kd, fxc, hfxc);
            }
          }
        } else if (t_coef) {
          const float tx_ = __builtin_truncf(psx), ty_ = __builtin_truncf(psy);
          const float fx = psx - tx_, fy = psy - ty_;
          const float hfx = 0.5f * fx, hfy = 0.5f * fy;
          // one 24-bit multiply per pixel (a 32-bit v_mul_lo_u32 issues at quarter rate); every other term
          // of the two addresses is wave-uniform and folded into tap_base / c_delta when the block is planned
          const int tap = __mul24((int)ty_, cur.spitch()) + (int)tx_ + cur.tap_base; // window slot of (int(sx) - 1, int(sy)): the second tap row
          const float4 *tb = win + tap;
          const float4 *ci = win + (tap + cur.c_delta(h));
          const float4 *cm = ci + cur.c_plane, *cc = cm + cur.c_plane;
          // the only reads of the raw window: the second tap row.  In the last pass they are the
          // block's last reads of it, and the next window's DMA goes right behind them
          const Rgba b0 = as_rgba(tb[0]), b1 = as_rgba(tb[1]), b2 = as_rgba(tb[2]), b3 = as_rgba(tb[3]);
          if (last_pass && CH != 5) next_window();
          auto vert = [&](int j, const Rgba bj) {
#if defined(LRP_SKIP_TAP_READS) // timing experiment (wrong results): no LDS reads of the planes
            const Rgba inner{f2{fx, fy} * (float)j, f2{hfx, fy}, 0.0f}, m0{f2{fy, fx} + (float)j, f2{fx, hfy}, 0.0f}, cma{f2{hfx, hfy}, f2{fy, fx} - (float)j, 0.0f};
            (void)ci; (void)cm; (void)cc;
#else
            const Rgba inner = as_rgba(ci[j]), m0 = as_rgba(cm[j]), cma = as_rgba(cc[j]);
#endif
            Rgba r = px_zero<4>();
            r.lo = bj.lo + hfy * (cma.lo + fy * (m0.lo + fy * inner.lo));
            r.hi = bj.hi + hfy * (cma.hi + fy * (m0.hi + fy * inner.hi));
            return r;
          };
#if LRP_VERT_STEPS
          // The four vertical evaluations step by step across the eight channel-pair chains instead of chain by chain:
          // a step's eight instructions are independent and the next step's operands are eight instructions old, so a
          // wavefront never waits on its own previous instruction (chain by chain, the compiler's order, every other
          // instruction depends on its predecessor and is preceded by an s_nop).  The plane reads are issued in the
          // order the steps consume them.  Same operations on the same operands.
          (void)vert;
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
#else
          const Rgba k0 = vert(0, b0), k1 = vert(1, b1), k2 = vert(2, b2), k3 = vert(3, b3);
#endif
          s = cubic4(k0, k1, k2, k3, fx, hfx);
          if constexpr (CH == 5) s.e = depth_from_window(win, tap - cur.spitch(), fx, fy, hfx, hfy, last_pass);
        } else if (t_staged) {
          // (split blocks: h = 1 reads the window of passes 2-3, requested behind pass 1's taps — its arithmetic and store and
          // the other wavefronts cover part of the round trip — and waited for in front of pass 2)
          const int half = (t_split && h == 1) ? 1 : 0;
          if (t_split && k == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); // the DMA is older than every store behind it
          const float tx_ = __builtin_truncf(psx), ty_ = __builtin_truncf(psy);
          const float fx = psx - tx_, fy = psy - ty_;
          const int slot0 = cur.org() + __mul24((int)ty_ - 1 - (kSplit ? cur.first_row_of(half) : cur.y_lo), cur.spitch()) + ((int)tx_ - 1 - cur.x_lo);
          const float4 *t = win + slot0;
          const float hfx = 0.5f * fx, hfy = 0.5f * fy;
          const float4 *t1 = t + cur.spitch(), *t2 = t1 + cur.spitch(), *t3 = t2 + cur.spitch();
          Rgba q[4][4]; // all 16 taps first: they are the last reads of the window in the last pass
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            q[j][0] = as_rgba(t[j]);
            q[j][1] = as_rgba(t1[j]);
            q[j][2] = as_rgba(t2[j]);
            q[j][3] = as_rgba(t3[j]);
          }
          if (last_pass && CH != 5) next_window();
          if (t_split && k == 1 && CH != 5) issue(P.src, cur, 1);
          const Rgba k0 = cubic4(q[0][0], q[0][1], q[0][2], q[0][3], fy, hfy);
          const Rgba k1 = cubic4(q[1][0], q[1][1], q[1][2], q[1][3], fy, hfy);
          const Rgba k2 = cubic4(q[2][0], q[2][1], q[2][2], q[2][3], fy, hfy);
          const Rgba k3 = cubic4(q[3][0], q[3][1], q[3][2], q[3][3], fy, hfy);
          s = cubic4(k0, k1, k2, k3, fx, hfx);
          if constexpr (CH == 5) {
            s.e = depth_from_window(win, slot0, fx, fy, hfx, hfy, last_pass, half);
            if (t_split && k == 1) issue(P.src, cur, 1); // behind the depth taps: pass 1's last reads of the window
          }
        } else if (kPassWin && P.win_split != 0 && pass_window(psx, psy, s, last_pass)) {
          // (rendered from the window of this pass)
        } else {
          if (last_pass) next_window(); // nothing staged: no tap of this block reads the window
          if constexpr (CH == 5) {
            // (colour and depth taps fetched one set after the other — 16 dwordx4 + 16 dword loads instead of 20 dwordx4, at
            // three or at four waves per SIMD — is 18-26 % slower: this path is bound by the number of gather instructions)
            const Px<5> s5 = sample_direct<2, Loop, 5, LRP_WIN_MINWAVES5 >= 4>(P, src, psx, psy); // (LowReg at 128 VGPRs)
            s = Rgba{s5.lo, s5.hi, s5.e};
          } else {
            s = sample_direct<2, Loop, 4, (LRP_WIN_MINWAVES >= 5), 4 * CH>(P, src, psx, psy); // (LowReg when five waves per SIMD are asked for: 96 VGPRs)
          }
        }
        emit(g, k, s, std::integral_constant<bool, kRunsEverywhere>{}, !GeoRead || P.rgbaz_runs != 0);
      }
    }
    if (!dma_early && has_next()) issue_next(); // after the last read of the planes
    if (last_frame) cur = nxt;
   }
  }
}

using TileKernelFn = void (*)(const KParams);

// The GeoRead tile kernels: one per source mode.
template <int Interp, int CH> struct TileGeoKernelTable {
  static TileKernelFn get(int in_mode) {
    static_assert(Interp != 2, "nearest / bilinear");
    static const TileKernelFn table[4] = {
        reproject_tile_kernel<kRect, kInRect, Interp, CH, false, true>, reproject_tile_kernel<kRect, kInEquidistant, Interp, CH, false, true>,
        reproject_tile_kernel<kRect, kInEquirect, Interp, CH, false, true>, reproject_tile_kernel<kRect, kInEquirectLoop, Interp, CH, false, true>};
    return table[in_mode];
  }
};

template <int Interp, int CH, bool Frames> struct TileKernelTable {
  static TileKernelFn get(int out_idx, int in_mode) {
    static const TileKernelFn table[3][4] = {
        {reproject_tile_kernel<kRect, kInRect, Interp, CH, Frames>, reproject_tile_kernel<kRect, kInEquidistant, Interp, CH, Frames>,
         reproject_tile_kernel<kRect, kInEquirect, Interp, CH, Frames>, reproject_tile_kernel<kRect, kInEquirectLoop, Interp, CH, Frames>},
        {reproject_tile_kernel<kEquidistant, kInRect, Interp, CH, Frames>,
         reproject_tile_kernel<kEquidistant, kInEquidistant, Interp, CH, Frames>,
         reproject_tile_kernel<kEquidistant, kInEquirect, Interp, CH, Frames>,
         reproject_tile_kernel<kEquidistant, kInEquirectLoop, Interp, CH, Frames>},
        {reproject_tile_kernel<kEquirect, kInRect, Interp, CH, Frames>, reproject_tile_kernel<kEquirect, kInEquidistant, Interp, CH, Frames>,
         reproject_tile_kernel<kEquirect, kInEquirect, Interp, CH, Frames>,
         reproject_tile_kernel<kEquirect, kInEquirectLoop, Interp, CH, Frames>}};
    return table[out_idx][in_mode];
  }
};

// P.channels must be 3, 4 or 5.
template <int Interp> hipError_t launch_tile_interp(KParams P, int out_idx, int in_mode, hipStream_t stream) {
  constexpr int tile_h = tile_rows<Interp>() * kT2Waves;
  if (P.quad) { // the top-left quadrant only: every pixel also renders its three mirror images
    P.tiles_x = ((P.out_w + 1) / 2 + kT2W - 1) / kT2W;
    P.tiles_y = ((P.out_h + 1) / 2 + tile_h - 1) / tile_h;
  } else {
    P.tiles_x = (P.out_w + kT2W - 1) / kT2W;
    const int rows = P.y_end - P.y_offset;
    P.tiles_y = (rows + tile_h - 1) / tile_h;
  }
  const int n_tiles = P.tiles_x * P.tiles_y;
  if (n_tiles <= 0) return hipSuccess;
  // Frames per wavefront of a batched launch (nearest / bilinear, one sample per pixel): as many as leave at least two
  // rounds of workgroups on the chip.
  int groups = P.batch_n > 0 ? P.batch_n : 1;
  const int frames_override = P.frames_per_wave; // on entry: 0 = automatic
  P.frames_per_wave = 1;
  if (Interp != 2 && P.num_samples == 1 && P.batch_n > 1 && P.quad == 0) { // (the plain path: any rotation; the mirrored paths are bound by memory)
    const long long units = (long long)n_tiles * P.batch_n;
    int F = (int)std::min<long long>(P.batch_n, std::max<long long>(1, units / 4096));
    if (out_idx == 2 && in_mode == kInRect) F = 1; // (see the window kernel: uneven tiles)
    if (frames_override > 0) F = std::max(1, std::min(P.batch_n, frames_override)); // the caller's override (lrp_debug_set "batch_frames": A/B runs, tests)
    P.frames_per_wave = F;
    groups = (P.batch_n + F - 1) / F;
  }
  TileKernelFn fn;
  if (P.geo_mode == 2) { // coordinates from the geometry cache (the host asks for it for single whole-image launches only)
    if constexpr (Interp != 2) {
      if (P.quad != 0 || P.batch_n > 0 || P.num_samples != 1 || P.y_offset != 0 || P.y_end != P.out_h) return hipErrorInvalidValue;
      fn = P.channels == 4 ? TileGeoKernelTable<Interp, 4>::get(in_mode) : P.channels == 3 ? TileGeoKernelTable<Interp, 3>::get(in_mode) : TileGeoKernelTable<Interp, 5>::get(in_mode);
    } else {
      return hipErrorInvalidValue;
    }
  } else if (P.frames_per_wave > 1) {
    if constexpr (Interp != 2)
      fn = P.channels == 4   ? TileKernelTable<Interp, 4, true>::get(out_idx, in_mode)
           : P.channels == 3 ? TileKernelTable<Interp, 3, true>::get(out_idx, in_mode)
                             : TileKernelTable<Interp, 5, true>::get(out_idx, in_mode);
    else
      return hipErrorInvalidValue;
  } else {
    fn = P.channels == 4   ? TileKernelTable<Interp, 4, false>::get(out_idx, in_mode)
         : P.channels == 3 ? TileKernelTable<Interp, 3, false>::get(out_idx, in_mode)
                           : TileKernelTable<Interp, 5, false>::get(out_idx, in_mode);
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)(kXcds * xcd_rows(P.tiles_y) * P.tiles_x), (unsigned)groups), dim3(kT2Threads), 0, stream, P);
  return hipGetLastError();
}

// The window kernel of one (output lens, source mode) cell for a mirror mode, or null where the mode does not exist
// (rows-only needs the column-separable source x: no equidistant lens on either side; columns-only a rectilinear target).
template <int OutLens, int InMode, int QMode, int CH, bool Frames> constexpr TileKernelFn win_kernel_fn() {
  if constexpr (QMode == 2 && (OutLens == kEquidistant || InMode == kInEquidistant))
    return nullptr;
  else if constexpr (QMode == 3 && OutLens != kRect)
    return nullptr;
  else if constexpr (QMode == 4 && OutLens != kEquidistant)
    return nullptr;
  else
    return reproject_bicubic_win_kernel<OutLens, InMode, QMode, CH, Frames>;
}
template <int QMode, int CH, bool Frames> struct WinKernelTable {
  static TileKernelFn get(int out_idx, int in_mode) {
    static const TileKernelFn table[3][4] = {
        {win_kernel_fn<kRect, kInRect, QMode, CH, Frames>(), win_kernel_fn<kRect, kInEquidistant, QMode, CH, Frames>(),
         win_kernel_fn<kRect, kInEquirect, QMode, CH, Frames>(), win_kernel_fn<kRect, kInEquirectLoop, QMode, CH, Frames>()},
        {win_kernel_fn<kEquidistant, kInRect, QMode, CH, Frames>(), win_kernel_fn<kEquidistant, kInEquidistant, QMode, CH, Frames>(),
         win_kernel_fn<kEquidistant, kInEquirect, QMode, CH, Frames>(), win_kernel_fn<kEquidistant, kInEquirectLoop, QMode, CH, Frames>()},
        {win_kernel_fn<kEquirect, kInRect, QMode, CH, Frames>(), win_kernel_fn<kEquirect, kInEquidistant, QMode, CH, Frames>(),
         win_kernel_fn<kEquirect, kInEquirect, QMode, CH, Frames>(), win_kernel_fn<kEquirect, kInEquirectLoop, QMode, CH, Frames>()}};
    return table[out_idx][in_mode];
  }
};

// The GeoRead instantiations (plain blocks, coordinates from the geometry cache): one per source mode.
// ... and with the frame loop: batched launches of a geometry whose entry exists.  A wavefront loads the coordinates and the
// extremes of its block once and renders it for up to 16 frames (same box, 16-frame launches: headline 104.3 -> 100.9 us per
// frame, general rotation 98.3 -> 95.7 against the instantiations that compute their coordinates once per 16 frames — those
// carry the lens math in registers: 68-90 spilled SGPRs against 11-22 here).
template <int CH> struct WinGeoFramesKernelTable {
  static TileKernelFn get(int in_mode) {
    static const TileKernelFn table[4] = {
        reproject_bicubic_win_kernel<kRect, kInRect, 0, CH, true, true>, reproject_bicubic_win_kernel<kRect, kInEquidistant, 0, CH, true, true>,
        reproject_bicubic_win_kernel<kRect, kInEquirect, 0, CH, true, true>, reproject_bicubic_win_kernel<kRect, kInEquirectLoop, 0, CH, true, true>};
    return table[in_mode];
  }
};
template <int CH> struct WinGeoKernelTable {
  static TileKernelFn get(int in_mode, bool big_windows) {
    static const TileKernelFn table[4] = {
        reproject_bicubic_win_kernel<kRect, kInRect, 0, CH, false, true>, reproject_bicubic_win_kernel<kRect, kInEquidistant, 0, CH, false, true>,
        reproject_bicubic_win_kernel<kRect, kInEquirect, 0, CH, false, true>, reproject_bicubic_win_kernel<kRect, kInEquirectLoop, 0, CH, false, true>};
    if (big_windows && in_mode == kInRect) return reproject_bicubic_win_kernel<kEquirect, kInRect, 0, CH, false, true>;
    return table[in_mode];
  }
};

// num_samples must be 1 (the pipeline keeps no accumulator across sub-samples).  QMode != 0: P.win_mode == QMode,
// set by the host only for cells where the mode exists.  GeoRead: P.geo_mode == 2, a single whole-image launch.
template <int QMode, int CH, bool GeoRead = false>
inline hipError_t launch_win_bicubic_impl(KParams P, int out_idx, int in_mode, hipStream_t stream) {
  static_assert(!GeoRead || QMode == 0, "the geometry cache feeds plain blocks");
  if (GeoRead && (P.geo_mode != 2 || P.y_offset != 0 || P.y_end != P.out_h)) return hipErrorInvalidValue;
  const int rows = P.y_end - P.y_offset;
  if (QMode != 0) {
    // the launch enumerates the top-left quadrant (the top / the left half when one axis is mirrored); a wavefront
    // renders a block and its mirror images
    const int qw = (QMode == 1 || QMode == 3 || QMode == 4) ? (P.out_w + 1) / 2 : P.out_w;
    const int qh = (QMode == 1 || QMode == 2 || QMode == 4) ? (P.out_h + 1) / 2 : P.out_h;
    P.tiles_x = (qw + kBlkW * kWinWaves - 1) / (kBlkW * kWinWaves);
    P.tiles_y = (qh + kBlkH - 1) / kBlkH;
    P.blocks_per_wave = (QMode == 1 || QMode == 4) ? 4 : 2;
  } else {
    P.tiles_x = (P.out_w + kBlkW * kWinWaves - 1) / (kBlkW * kWinWaves);
    // strips of LRP_WIN_STRIP blocks when that still leaves >= 8 workgroups per CU, else shorter
    const int row_blocks = (rows + kBlkH - 1) / kBlkH;
    int G = LRP_WIN_STRIP;
    if (GeoRead && P.blocks_per_wave > 0) G = std::min(P.blocks_per_wave, kGeoStripRows); // the caller's override (lrp_debug_set "geo_strip")
    // (a batch whose wavefronts walk several frames pipelines the windows of one block across its frames: one block per
    // wavefront measured 2-3 % faster there — equirect -> fisheye rotated 143 -> 139 us, rect -> rect 130.5 -> 128 —, four 5 % slower)
    if (P.batch_n > 1 && !(out_idx == 2 && in_mode == kInRect)) G = 1;
    const bool strip_forced = GeoRead && P.blocks_per_wave > 0;
    while (!strip_forced && G > 1 && (long long)P.tiles_x * kWinWaves * ((row_blocks + G - 1) / G) < 8192) G >>= 1; // >= 2 rounds of wavefronts
    P.blocks_per_wave = G;
    P.tiles_y = (row_blocks + G - 1) / G;
  }
  const int n_tiles = P.tiles_x * P.tiles_y;
  if (n_tiles <= 0) return hipSuccess;
  // Frames per wavefront of a batched launch: as many as leave at least two rounds of wavefronts on the chip
  // (4096 wave slots), so that a 4K batch of 16 runs every strip through all 16 frames and small images keep the chip full.
  int groups = P.batch_n > 0 ? P.batch_n : 1;
  const int frames_override = P.frames_per_wave; // on entry: 0 = automatic
  P.frames_per_wave = 1;
  if (P.batch_n > 1) {
    const long long units = (long long)n_tiles * kWinWaves * P.batch_n;
    int F = (int)std::min<long long>(P.batch_n, std::max<long long>(1, units / 8192));
    // a rectilinear view inside a panorama: a quarter of the strips (the ones in view) carry most of the frame's time and
    // gain nothing from shared coordinates (they wait for gathers) — 16 frames long they unbalance the launch (223 -> 256 us)
    if (out_idx == 2 && in_mode == kInRect) F = 1;
    if (frames_override > 0) F = std::max(1, std::min(P.batch_n, frames_override)); // the caller's override (lrp_debug_set "batch_frames": A/B runs, tests)
    if (P.geo_mode == 1 || P.geo_mode == 3) F = 1; // the launch that writes a geometry-cache entry: the instantiations without the frame loop have the side output
    P.frames_per_wave = F;
    groups = (P.batch_n + F - 1) / F;
  }
  TileKernelFn fn;
  if constexpr (GeoRead)
    fn = P.frames_per_wave > 1 ? WinGeoFramesKernelTable<CH>::get(in_mode) : WinGeoKernelTable<CH>::get(in_mode, P.rgbaz_runs != 0);
  else
    fn = P.frames_per_wave > 1 ? WinKernelTable<QMode, CH, true>::get(out_idx, in_mode) : WinKernelTable<QMode, CH, false>::get(out_idx, in_mode);
  if (!fn) return hipErrorInvalidValue; // (the host never asks for a mode outside its cells)
  hipLaunchKernelGGL(fn, dim3((unsigned)(kXcds * xcd_rows(P.tiles_y, kWinXcdBand) * P.tiles_x), (unsigned)groups), dim3(kWinThreads), 0, stream, P);
  return hipGetLastError();
}

} // namespace lrp
