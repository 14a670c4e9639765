// lrp_corner_fill.h — the corner runs of a geometry-cache entry (lrp_params.h "Block lists") written as row segments: every
// pixel of a run is the one value sample_bicubic gives when all 16 taps clamp to the same corner texel
// (src/reproject.cpp:109-148 with the clamped indices of :114-127 and weights of :130-131), finished like any other pixel
// (:334-341, fused post_process :421-437).  Used by the stand-alone fill kernel (lrp_geo_lists.hip) and, as a few store
// instructions at the end of every wavefront's life, by the window kernels of a listed launch (lrp_win_kernel.h).
#pragma once

#include "lrp_kernel_common.h"

namespace lrp {

// A row segment is contiguous in the interleaved output (src/reproject.cpp:49-51 layout); a wavefront writes it 1 KiB per
// instruction, lane i the 16 bytes at 16 i: for RGBA that is the pixel itself, for RGBAZ / RGB the repeating 5- / 3-float
// pattern at the phase the lane's bytes start at (the segment starts on a pixel, 1 KiB further the phase has advanced by
// 256 mod 5 = 1 / 256 mod 3 = 1 floats: one pre-rotated vector per instruction of a row, the same for every row).
template <int CH> constexpr int corner_fill_vectors() { return (kGeoRunBlocks * 16 * CH * 4 + 1023) / 1024; } // instructions per full row segment

// Row segments [seg_first, seg_end) of the runs (segment s = pixel row s % 16 of run s / 16) of one frame: `src` supplies the
// corner texels, `dst` receives the rows.  seg_first / seg_end are wave-uniform; all 64 lanes of the wavefront take part.
template <int CH>
__device__ __forceinline__ void corner_fill_rows(const KParams &P, const float *src, float *dst, uint32_t seg_first, uint32_t seg_end) {
  static_assert(CH == 3 || CH == 4 || CH == 5, "RGB, RGBA or RGBAZ");
  typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
  typedef const uint32_t __attribute__((address_space(4))) *ScalarU;
  typedef const float __attribute__((address_space(4))) *ScalarF;
  const int lane = (int)(threadIdx.x & 63u);
  auto uniform_ptr = [](const void *p, uint32_t byte_offset) {
    const uintptr_t base = reinterpret_cast<uintptr_t>(p);
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32 |
            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base)) + byte_offset;
  };
  constexpr int kVec = corner_fill_vectors<CH>();
  v4f_a4 q[kVec];
#pragma unroll
  for (int i = 0; i < kVec; ++i) q[i] = v4f_a4{0.0f, 0.0f, 0.0f, 0.0f};
  int cur_run = -1, cur_cls = 0, row_blk = 0, x_first = 0, seg_floats = 0;
#pragma unroll 1
  for (uint32_t s = seg_first; s < seg_end; ++s) {
    const int run_index = (int)(s >> 4), r = (int)(s & 15u);
    if (run_index != cur_run) {
      cur_run = run_index;
      // (< 2^28 runs: 16 bytes each)
      const ScalarU run = reinterpret_cast<ScalarU>(uniform_ptr(P.geo_runs, (uint32_t)__builtin_amdgcn_readfirstlane(run_index * 16)));
      row_blk = (int)run[0];
      x_first = (int)run[1] * 16;
      seg_floats = (min(x_first + (int)run[2] * 16, P.out_w) - x_first) * CH;
      const int cls = (int)run[3];
      if (cls != cur_cls) {
        cur_cls = cls;
        // the one value: sample_bicubic with all 16 taps on the corner texel — the same operations as the window kernel's
        // corner_value (lrp_win_kernel.h) and sample_direct's one-column-and-one-row case
        const int xh = (cls - 1) & 1, yh = (cls - 1) >> 1;
        const float fx = xh ? 1.0f : 0.0f, fy = yh ? 1.0f : 0.0f; // the clamped weights (src/reproject.cpp:130-131)
        const float hfx = 0.5f * fx, hfy = 0.5f * fy;
        const uint32_t row_bytes = (uint32_t)P.in_w * (4u * CH);
        const uint32_t off = (uint32_t)(yh ? P.in_h - 1 : 0) * row_bytes + (uint32_t)(xh ? P.in_w - 1 : 0) * (4u * CH);
        const ScalarF tp = reinterpret_cast<ScalarF>(uniform_ptr(src, (uint32_t)__builtin_amdgcn_readfirstlane((int)off)));
        Px<CH> a; // src/reproject.cpp:334-336 with num_samples == 1: 0.0f + sample
        if constexpr (CH == 5) {
          const Px<5> t{f2{tp[0], tp[1]}, f2{tp[2], tp[3]}, tp[4]};
          const Px<5> kv = cubic_px<5>(t, t, t, t, fy, hfy);
          const Px<5> rv = cubic_px<5>(kv, kv, kv, kv, fx, hfx);
          Rgba a4 = px_zero<4>();
          px_add<4>(a4, Rgba{rv.lo, rv.hi, 0.0f});
          a = Px<5>{a4.lo, a4.hi, 0.0f + rv.e};
        } else {
          const Px<4> t{f2{tp[0], tp[1]}, f2{tp[2], CH == 3 ? 0.0f : tp[3]}, 0.0f};
          const Px<4> kv = cubic_px<4>(t, t, t, t, fy, hfy);
          const Px<4> rv = cubic_px<4>(kv, kv, kv, kv, fx, hfx);
          Rgba a4 = px_zero<4>();
          px_add<4>(a4, rv);
          a = Px<CH>{a4.lo, CH >= 4 ? a4.hi : f2{0.0f, 0.0f}, CH == 3 ? a4.hi.x : 0.0f};
        }
        // :338-341 with normalize == 1 (x * 1.0f is x for every float: finish_px<CH, true>) and the fused post_process
        float c[5] = {a.lo.x, a.lo.y, CH == 3 ? a.e : a.hi.x, a.hi.y, a.e};
        if (P.has_post != 0) {
          c[0] = tonemap(c[0], P.exposure, P.reinhard);
          c[1] = tonemap(c[1], P.exposure, P.reinhard);
          c[2] = tonemap(c[2], P.exposure, P.reinhard);
        }
        // the vector of instruction i of a row: floats 4 lane + 256 i .. + 3 of the repeating pattern c[0] .. c[CH - 1]
#pragma unroll
        for (int i = 0; i < kVec; ++i) {
          const int m = (4 * lane + 256 * i) % CH;
          auto at = [&](int n) {
            const int j = (m + n) % CH;
            float v = c[0];
#pragma unroll
            for (int t = 1; t < CH; ++t) v = j == t ? c[t] : v;
            return v;
          };
          q[i] = v4f_a4{at(0), at(1), at(2), at(3)};
        }
      }
    }
    const int y = row_blk * 16 + r;
    if (y >= P.out_h) continue; // (the last row of blocks of an image whose height is not a multiple of 16)
    float *const seg = dst + ((size_t)y * (size_t)P.out_w + (size_t)x_first) * CH;
#pragma unroll
    for (int i = 0; i < kVec; ++i) {
      const int f0 = 4 * lane + 256 * i; // first float of this lane's vector
      if (f0 + 4 <= seg_floats) {
        __builtin_nontemporal_store(q[i], reinterpret_cast<v4f_a4 *>(seg + f0));
      } else if (f0 < seg_floats) { // the last, partial vector of a segment that the image's right edge cut (RGB / RGBAZ)
        __builtin_nontemporal_store(q[i].x, seg + f0);
        if (f0 + 1 < seg_floats) __builtin_nontemporal_store(q[i].y, seg + f0 + 1);
        if (f0 + 2 < seg_floats) __builtin_nontemporal_store(q[i].z, seg + f0 + 2);
      }
    }
  }
}

} // namespace lrp
