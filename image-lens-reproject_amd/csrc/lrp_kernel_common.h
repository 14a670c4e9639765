// lrp_kernel_common.h — what the tile and the window kernels share (lrp_kernel_v2.h is the umbrella): wave-wide reductions,
// the coordinate pipeline with hoisted lens terms (output pixel -> ray -> rotation -> source lens -> texel coordinates),
// texels as packed channel pairs, Catmull-Rom, taps through a buffer descriptor, the interior votes, the per-pixel
// samplers (sample_direct) and the output stores.
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
typedef float f2 __attribute__((ext_vector_type(2)));
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
  const f2 mid = ((((2.0f * a) - (5.0f * b)) + (4.0f * c)) - d) + t * inner;
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

// ---- supersampling with a lane per sub-sample: the ordered sum of a pixel ------------------
// The n = ns^2 sub-samples of a pixel sit in n consecutive lanes in the reference's order (sub = ns ssx + ssy); `first`: this
// lane holds the pixel's first one.  The reference sums them into a zero-initialised accumulator in that order
// (src/reproject.cpp:334-336: 0.0f + s0 + s1 + ...).  A chain of n - 1 steps: a first lane starts from 0.0f + s, every other
// from s (a placeholder); step t replaces every lane's value by (its left neighbour's value) + s — after it the lane of
// sub-sample t holds the reference's sum up to and including st, whatever the lanes of later sub-samples hold meanwhile; the
// pixel's last lane ends up with its sum.  One VOP2 add with a DPP operand (wave_shr:1) per component and step (as a move +
// a packed add it is six instructions per step instead of four: num_samples 4 runs fifteen steps).  A DPP operand written
// by the previous VALU instruction needs two wait states: the s_nop in front of a step's first add; the step's other adds
// keep a component's add of step t + 1 at least two instructions behind its add of step t.  All 64 lanes must be active.
template <int CH> __device__ __forceinline__ Px<CH> ss_ordered_sum(const Px<CH> &sp, bool first, int n) {
  Px<CH> a = sp;
  if (first) { // 0.0f + s: turns -0 into +0 and quiets a NaN like the reference's first += does
    a = px_zero<CH>();
    px_add<CH>(a, sp);
  }
  float a0 = a.lo.x, a1 = a.lo.y, a2 = CH >= 4 ? a.hi.x : a.e, a3 = CH >= 4 ? a.hi.y : 0.0f, a4 = CH == 5 ? a.e : 0.0f;
  const float s0 = sp.lo.x, s1 = sp.lo.y, s2 = CH >= 4 ? sp.hi.x : sp.e, s3 = CH >= 4 ? sp.hi.y : 0.0f, s4 = CH == 5 ? sp.e : 0.0f;
#pragma unroll 1
  for (int t = 1; t < n; ++t) {
    if constexpr (CH == 3)
      asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %1, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %5 wave_shr:1 row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1), "+v"(a2)
                   : "v"(s0), "v"(s1), "v"(s2));
    else if constexpr (CH == 4)
      asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %1, %1, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %3, %3, %7 wave_shr:1 row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                   : "v"(s0), "v"(s1), "v"(s2), "v"(s3));
    else
      asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %1, %1, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %3, %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %4, %4, %9 wave_shr:1 row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4)
                   : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4));
  }
  a.lo = f2{a0, a1};
  if constexpr (CH >= 4) a.hi = f2{a2, a3};
  if constexpr (CH == 3) a.e = a2;
  if constexpr (CH == 5) a.e = a4;
  return a;
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
      s = bicubic_taps<CH, true, LowReg>(rsrc, v0, v0 + T, v0 + 2u * T, v0 + 3u * T, 0u, row_bytes, 2u * row_bytes,
                                 3u * row_bytes, fx, fy);
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
  asm volatile("" : "+s"(has_post));
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

} // namespace lrp
