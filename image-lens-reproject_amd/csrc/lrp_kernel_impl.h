// lrp_kernel_impl.h — the reprojection kernel template and its launch table.
// Included by one .hip translation unit per interpolation mode so the three
// instantiation sets compile in parallel.
//
// Mapping (gfx950): a 256-thread workgroup owns a TILE_W x TILE_H block of
// output pixels; each 64-lane wavefront owns two contiguous 32-pixel row
// segments of it, one pixel per lane, so a wavefront's RGBA stores are two
// contiguous 512-byte runs and its source taps stay inside a compact window
// (L1/L2 friendly).  Tiles are handed out so that the workgroups the dispatcher
// places on one XCD (blockIdx % 8 equal) walk bands of neighbouring tile rows
// (xcd_tile(), lrp_params.h): neighbouring tiles share source rows, which then
// hit in that XCD's private 4 MiB L2.
#pragma once

#include "lrp_device.h"

namespace lrp {

constexpr int kTileW = 32;
constexpr int kTileH = 8;
constexpr int kThreads = kTileW * kTileH; // 256 = 4 wavefronts

template <int OutLens, int InMode, int Interp, int CH>
__global__ __launch_bounds__(kThreads) void reproject_kernel(const KParams P) {
  constexpr bool Loop = (InMode == kInEquirectLoop);
  int tx, ty;
  if (!xcd_tile(P.tiles_x, P.tiles_y, tx, ty)) return;
  const int x = tx * kTileW + (int)(threadIdx.x % kTileW);
  const int y = P.y_offset + ty * kTileH + (int)(threadIdx.x / kTileW);
  if (x >= P.out_w || y >= P.y_end) return;

  // pixel centre in image-centred coordinates (src/reproject.cpp:287-288)
  const float cx = ((float)x + 0.5f) - (float)P.out_w * 0.5f;
  const float cy = ((float)y + 0.5f) - (float)P.out_h * 0.5f;

  Texel<CH> acc;
#pragma unroll
  for (int c = 0; c < texel_lanes<CH>(); ++c) acc.v[c] = 0.0f;

  const int ns = P.num_samples;
  const float ns1 = (float)ns + 1.0f;
  for (int ssx = 0; ssx < ns; ++ssx) {
    const float scx = cx + ((float)ssx + 1.0f) / ns1 - 0.5f; // src/reproject.cpp:295
    for (int ssy = 0; ssy < ns; ++ssy) {
      const float scy = cy + ((float)ssy + 1.0f) / ns1 - 0.5f; // src/reproject.cpp:298
      float sx, sy;
      source_position<OutLens, InMode>(P, scx, scy, sx, sy);
      const Texel<CH> s = sample<Interp, CH, Loop>(P, sx, sy);
#pragma unroll
      for (int c = 0; c < texel_lanes<CH>(); ++c) acc.v[c] += s.v[c]; // :334-336
    }
  }
  // src/reproject.cpp:338-341 (the last store is the one that survives)
#pragma unroll
  for (int c = 0; c < texel_lanes<CH>(); ++c) acc.v[c] = acc.v[c] * P.normalize;
  if (P.has_post) {
    // fused post_process: first min(C,3) channels (src/reproject.cpp:423-434)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (c < texel_lanes<CH>() && c < P.ch_count) acc.v[c] = tonemap(acc.v[c], P.exposure, P.reinhard);
  }
  const uint32_t off = ((uint32_t)y * (uint32_t)P.out_w + (uint32_t)x) * (uint32_t)P.channels;
  store_texel<CH>(P.dst, off, acc, P.ch_count);
}

using KernelFn = void (*)(const KParams);

template <int Interp, int CH> struct KernelTable {
  // [out lens 0..2][in mode 0..3]
  static KernelFn get(int out_idx, int in_mode) {
    static const KernelFn table[3][4] = {
        {reproject_kernel<kRect, kInRect, Interp, CH>, reproject_kernel<kRect, kInEquidistant, Interp, CH>,
         reproject_kernel<kRect, kInEquirect, Interp, CH>, reproject_kernel<kRect, kInEquirectLoop, Interp, CH>},
        {reproject_kernel<kEquidistant, kInRect, Interp, CH>,
         reproject_kernel<kEquidistant, kInEquidistant, Interp, CH>,
         reproject_kernel<kEquidistant, kInEquirect, Interp, CH>,
         reproject_kernel<kEquidistant, kInEquirectLoop, Interp, CH>},
        {reproject_kernel<kEquirect, kInRect, Interp, CH>, reproject_kernel<kEquirect, kInEquidistant, Interp, CH>,
         reproject_kernel<kEquirect, kInEquirect, Interp, CH>,
         reproject_kernel<kEquirect, kInEquirectLoop, Interp, CH>}};
    return table[out_idx][in_mode];
  }
};

// out_idx: 0 rectilinear, 1 equidistant, 2 equirectangular.
template <int Interp>
hipError_t launch_interp(KParams P, int out_idx, int in_mode, hipStream_t stream) {
  P.tiles_x = (P.out_w + kTileW - 1) / kTileW;
  const int rows = P.y_end - P.y_offset;
  P.tiles_y = (rows + kTileH - 1) / kTileH;
  const int n_tiles = P.tiles_x * P.tiles_y;
  if (n_tiles <= 0) return hipSuccess;
  const dim3 grid((unsigned)(kXcds * xcd_rows(P.tiles_y) * P.tiles_x)), block(kThreads);
  KernelFn fn = (P.channels == 4) ? KernelTable<Interp, 4>::get(out_idx, in_mode)
                                  : KernelTable<Interp, 0>::get(out_idx, in_mode);
  hipLaunchKernelGGL(fn, grid, block, 0, stream, P);
  return hipGetLastError();
}

} // namespace lrp
