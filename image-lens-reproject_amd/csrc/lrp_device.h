// lrp_device.h — device-side lens models and samplers for gfx950.
//
// Every expression keeps the reference's association order and un-fused
// rounding (translation units are compiled with -ffp-contract=off); the
// file:line of the reference statement each one reproduces is cited.
#pragma once

#include <hip/hip_runtime.h>

#include "lrp_math.h"
#include "lrp_params.h"

namespace lrp {

// Workgroup -> tile (see kXcdBand in lrp_params.h); false: surplus workgroup.
template <int Band = kXcdBand> __device__ __forceinline__ bool xcd_tile(int tiles_x, int tiles_y, int &tx, int &ty, uint32_t workgroup = blockIdx.x) {
  const int xcd = (int)(workgroup % kXcds), j = (int)(workgroup / kXcds); // j: index within the XCD
  const int row = j / tiles_x; // row of tiles among this XCD's
  tx = j - row * tiles_x;
  ty = ((row / Band) * kXcds + xcd) * Band + row % Band;
  return ty < tiles_y;
}

// ---- scalar semantics inherited from the reference's x86-64 build ---------

// int(float) as cvttss2si executes it: truncation; NaN / inf / out of range
// give INT_MIN.  (v_cvt_i32_f32 alone would saturate and map NaN to 0.)
__device__ __forceinline__ int trunc_x86(float v) {
  int r = (int)v;
  return (__builtin_fabsf(v) < 2147483648.0f) ? r : (int)0x80000000;
}

// clamp<int> (src/reproject.cpp:33-35)
__device__ __forceinline__ int clamp_index(int x, int hi) {
  int m = (hi < x) ? hi : x;
  return (0 < m) ? m : 0;
}

// (i + W) % W of the wrapping samplers (src/reproject.cpp:43,60-61,114-117),
// two's-complement add and C remainder; a negative remainder (out-of-bounds
// read in the reference) is defined as column 0.  The two common cases avoid
// the integer division.
__device__ __forceinline__ int wrap_index(int i, int w) {
  int t = (int)((unsigned)i + (unsigned)w);
  if ((unsigned)t < (unsigned)w) return t;
  int t2 = t - w;
  if ((unsigned)t2 < (unsigned)w) return t2;
  // rare (coordinates beyond one image width outside the frame).  The divisor is made opaque so that the
  // reciprocal the division expands to is computed here, not hoisted into a VGPR that then lives — and
  // spills — across the whole kernel.
  int wv = w;
  asm volatile("" : "+v"(wv));
  int r = t % wv;
  return (r < 0) ? 0 : r;
}

template <bool Loop> __device__ __forceinline__ int column(int i, int w) {
  if constexpr (Loop)
    return wrap_index(i, w);
  else
    return clamp_index(i, w - 1);
}

// std::max(0.0f, std::min(1.0f, v)) with libstdc++'s comparison direction
// (src/reproject.cpp:70-71,130-131): NaN -> 1, -0 -> +0.
__device__ __forceinline__ float unit_clamp(float v) {
  float m = (v < 1.0f) ? v : 1.0f;
  return (0.0f < m) ? m : 0.0f;
}

// ---- output lens: pixel -> ray ---------------------------------------------

template <int OutLens>
__device__ __forceinline__ void target_ray(const LensP &L, float img_w, float img_h, float cx,
                                           float cy, float &vx, float &vy, float &vz) {
  if constexpr (OutLens == kRect) {
    // rectilinear_to_vec, src/reproject.cpp:152-158
    const float focal = L.p[0];
    vx = cx / img_w * L.sensor_width / focal;
    vy = cy / img_h * L.sensor_height / focal;
    vz = -1.0f;
  } else if constexpr (OutLens == kEquidistant) {
    // equidistant_to_vec, src/reproject.cpp:171-186
    const float fov = L.p[0];
    const float r_px = lrp_sqrtf(cx * cx + cy * cy);
    const float r_mm = r_px / img_w * L.sensor_width;
    const float focal = L.sensor_width / fov;
    const float theta = r_mm / focal;
    float sn, cs;
    sincosf_(theta, sn, cs);
    const float s = sn / r_px;
    vx = s * cx;
    vy = s * cy;
    vz = cs;
  } else {
    // equirectangular_to_vec, src/reproject.cpp:245-257
    const float lat_min = L.p[0], lat_max = L.p[1], lon_min = L.p[2], lon_max = L.p[3];
    const float lon_span = lon_max - lon_min;
    const float lat_span = lat_max - lat_min;
    const float lon = ((cx / img_w) + 0.5f) * lon_span + lon_min;
    const float lat = ((cy / img_h) + 0.5f) * lat_span + lat_min;
    float sn, cs;
    sincosf_(lon, sn, cs);
    vx = sn;
    vz = -cs;
    vy = sinf_(lat);
  }
}

// ---- input lens: ray -> centred source coordinates -------------------------

template <int InMode>
__device__ __forceinline__ void ray_to_source(const LensP &L, float img_w, float img_h, float x,
                                              float y, float z, float &cx, float &cy) {
  if constexpr (InMode == kInRect) {
    // vec_to_rectilinear, src/reproject.cpp:160-167
    const float focal = L.p[0];
    x = x / -z;
    y = y / -z;
    cx = x * img_w / L.sensor_width * focal;
    cy = y * img_h / L.sensor_height * focal;
  } else if constexpr (InMode == kInEquidistant) {
    // vec_to_equidistant, src/reproject.cpp:188-206
    const float fov = L.p[0];
    x = x / -z;
    y = y / -z;
    const float r = lrp_sqrtf(x * x + y * y);
    const float theta = atanf_(r);
    const float focal = L.sensor_width / fov;
    const float r_mm = focal * theta;
    const float r_px = r_mm / L.sensor_width * img_w;
    cx = x / r * r_px;
    cy = y / r * r_px;
  } else {
    // vec_to_equirectangular, src/reproject.cpp:259-271
    const float lat_min = L.p[0], lat_max = L.p[1], lon_min = L.p[2], lon_max = L.p[3];
    const float theta = -atan2f_(-x, -z);
    const float phi = asinf_(y / lrp_sqrtf(x * x + y * y + z * z));
    const float lon_span = lon_max - lon_min;
    const float lat_span = lat_max - lat_min;
    cx = ((theta - lon_min) / lon_span - 0.5f) * img_w;
    cy = ((phi - lat_min) / lat_span - 0.5f) * img_h;
  }
}

// One sub-sample position -> top-left-origin source texel coordinates
// (src/reproject.cpp:300-324).
template <int OutLens, int InMode>
__device__ __forceinline__ void source_position(const KParams &P, float scx, float scy, float &sx,
                                                float &sy) {
  float vx, vy, vz;
  target_ray<OutLens>(P.out_lens, (float)P.out_w, (float)P.out_h, scx, scy, vx, vy, vz);
  if (P.has_rot) {
    const float nx = P.rot[0] * vx + P.rot[1] * vy + P.rot[2] * vz;
    const float ny = P.rot[3] * vx + P.rot[4] * vy + P.rot[5] * vz;
    const float nz = P.rot[6] * vx + P.rot[7] * vy + P.rot[8] * vz;
    vx = nx;
    vy = ny;
    vz = nz;
  }
  float px, py;
  ray_to_source<InMode>(P.in_lens, (float)P.in_w, (float)P.in_h, vx, vy, vz, px, py);
  sx = (px - 0.5f) + (float)P.in_w * 0.5f;
  sy = (py - 0.5f) + (float)P.in_h * 0.5f;
}

// ---- texel access -----------------------------------------------------------

// A texel of CH channels held in registers.  CH == 0 is the run-time channel
// count path (any C >= 1, at most kMaxDynChannels).
constexpr int kMaxDynChannels = 8;

template <int CH> struct Texel {
  float v[CH == 0 ? kMaxDynChannels : CH];
};

template <int CH>
__device__ __forceinline__ Texel<CH> load_texel(const float *__restrict__ src, uint32_t elem_off, int ch) {
  Texel<CH> t;
  if constexpr (CH == 4) {
    const float4 q = *reinterpret_cast<const float4 *>(src + elem_off);
    t.v[0] = q.x;
    t.v[1] = q.y;
    t.v[2] = q.z;
    t.v[3] = q.w;
  } else if constexpr (CH == 0) {
#pragma unroll
    for (int c = 0; c < kMaxDynChannels; ++c)
      if (c < ch) t.v[c] = src[elem_off + c];
  } else {
#pragma unroll
    for (int c = 0; c < CH; ++c) t.v[c] = src[elem_off + c];
  }
  return t;
}

template <int CH> __device__ __forceinline__ void store_texel(float *__restrict__ dst, uint32_t elem_off, const Texel<CH> &t, int ch) {
  // non-temporal: the output is written once and never read back by the kernel
  if constexpr (CH == 4) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(v4f{t.v[0], t.v[1], t.v[2], t.v[3]}, reinterpret_cast<v4f *>(dst + elem_off));
  } else if constexpr (CH == 0) {
#pragma unroll
    for (int c = 0; c < kMaxDynChannels; ++c)
      if (c < ch) __builtin_nontemporal_store(t.v[c], dst + elem_off + c);
  } else {
#pragma unroll
    for (int c = 0; c < CH; ++c) __builtin_nontemporal_store(t.v[c], dst + elem_off + c);
  }
}

template <int CH> constexpr int texel_lanes() { return CH == 0 ? kMaxDynChannels : CH; }

// cubicInterpolate (src/reproject.cpp:92-98) in the reference's association
// order; `half_t` is 0.5f * t (computed once per axis: same bits).
__device__ __forceinline__ float catmull_rom(float a, float b, float c, float d, float t, float half_t) {
  const float inner = ((3.0f * (b - c)) + d) - a;
  const float mid = ((((2.0f * a) - (5.0f * b)) + (4.0f * c)) - d) + t * inner;
  const float outer = (c - a) + t * mid;
  return b + half_t * outer;
}

// ---- samplers ---------------------------------------------------------------

// sample_nearest, src/reproject.cpp:39-53
template <int CH, bool Loop>
__device__ __forceinline__ Texel<CH> sample_nearest(const KParams &P, float sx, float sy) {
  const int lx = column<Loop>(trunc_x86(sx + 0.5f), P.in_w);
  const int ly = clamp_index(trunc_x86(sy + 0.5f), P.in_h - 1);
  const uint32_t off = ((uint32_t)ly * (uint32_t)P.in_w + (uint32_t)lx) * (uint32_t)P.channels;
  return load_texel<CH>(P.src, off, P.ch_count);
}

// sample_bilinear, src/reproject.cpp:55-90
template <int CH, bool Loop>
__device__ __forceinline__ Texel<CH> sample_bilinear(const KParams &P, float sx, float sy) {
  const int w = P.in_w, h = P.in_h, C = P.channels;
  const int lx = column<Loop>(trunc_x86(sx), w);
  const int ux = column<Loop>(trunc_x86(sx + 1.0f), w);
  const int ly = clamp_index(trunc_x86(sy), h - 1);
  const int uy = clamp_index(trunc_x86(sy + 1.0f), h - 1);
  const float fx = unit_clamp(sx - (float)lx);
  const float fy = unit_clamp(sy - (float)ly);
  const float cfx = 1.0f - fx;
  const float cfy = 1.0f - fy;
  const uint32_t row_l = (uint32_t)ly * (uint32_t)w, row_u = (uint32_t)uy * (uint32_t)w;
  const Texel<CH> ll = load_texel<CH>(P.src, (row_l + lx) * C, P.ch_count);
  const Texel<CH> lu = load_texel<CH>(P.src, (row_l + ux) * C, P.ch_count);
  const Texel<CH> ul = load_texel<CH>(P.src, (row_u + lx) * C, P.ch_count);
  const Texel<CH> uu = load_texel<CH>(P.src, (row_u + ux) * C, P.ch_count);
  Texel<CH> r;
#pragma unroll
  for (int c = 0; c < texel_lanes<CH>(); ++c) {
    const float lo = fx * lu.v[c] + cfx * ll.v[c];
    const float hi = fx * uu.v[c] + cfx * ul.v[c];
    r.v[c] = fy * hi + cfy * lo;
  }
  return r;
}

// sample_bicubic + bicubicInterpolate, src/reproject.cpp:100-148: vertical
// cubic per tap column, then the horizontal cubic.
template <int CH, bool Loop>
__device__ __forceinline__ Texel<CH> sample_bicubic(const KParams &P, float sx, float sy) {
  const int w = P.in_w, h = P.in_h, C = P.channels;
  int xs[4], ys[4];
  xs[0] = column<Loop>(trunc_x86(sx - 1.0f), w);
  xs[1] = column<Loop>(trunc_x86(sx), w);
  xs[2] = column<Loop>(trunc_x86(sx + 1.0f), w);
  xs[3] = column<Loop>(trunc_x86(sx + 2.0f), w);
  ys[0] = clamp_index(trunc_x86(sy - 1.0f), h - 1);
  ys[1] = clamp_index(trunc_x86(sy), h - 1);
  ys[2] = clamp_index(trunc_x86(sy + 1.0f), h - 1);
  ys[3] = clamp_index(trunc_x86(sy + 2.0f), h - 1);
  const float fx = unit_clamp(sx - (float)xs[1]);
  const float fy = unit_clamp(sy - (float)ys[1]);
  const float hfx = 0.5f * fx, hfy = 0.5f * fy;
  uint32_t rows[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rows[j] = (uint32_t)ys[j] * (uint32_t)w;
  Texel<CH> col[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const Texel<CH> p0 = load_texel<CH>(P.src, (rows[0] + xs[i]) * C, P.ch_count);
    const Texel<CH> p1 = load_texel<CH>(P.src, (rows[1] + xs[i]) * C, P.ch_count);
    const Texel<CH> p2 = load_texel<CH>(P.src, (rows[2] + xs[i]) * C, P.ch_count);
    const Texel<CH> p3 = load_texel<CH>(P.src, (rows[3] + xs[i]) * C, P.ch_count);
#pragma unroll
    for (int c = 0; c < texel_lanes<CH>(); ++c)
      col[i].v[c] = catmull_rom(p0.v[c], p1.v[c], p2.v[c], p3.v[c], fy, hfy);
  }
  Texel<CH> r;
#pragma unroll
  for (int c = 0; c < texel_lanes<CH>(); ++c)
    r.v[c] = catmull_rom(col[0].v[c], col[1].v[c], col[2].v[c], col[3].v[c], fx, hfx);
  return r;
}

template <int Interp, int CH, bool Loop>
__device__ __forceinline__ Texel<CH> sample(const KParams &P, float sx, float sy) {
  if constexpr (Interp == 0)
    return sample_nearest<CH, Loop>(P, sx, sy);
  else if constexpr (Interp == 1)
    return sample_bilinear<CH, Loop>(P, sx, sy);
  else
    return sample_bicubic<CH, Loop>(P, sx, sy);
}

// post_process on one value (src/reproject.cpp:428-431)
__device__ __forceinline__ float tonemap(float v, float exposure, float reinhard) {
  v *= exposure;
  return v * (1.0f + v / (reinhard * reinhard)) / (1.0f + v);
}

} // namespace lrp
