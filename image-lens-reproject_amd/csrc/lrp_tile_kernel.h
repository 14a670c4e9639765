// lrp_tile_kernel.h — the tile kernel (nearest, bilinear, super-sampled / RGBAZ bicubic) and its launcher.
#pragma once

#include "lrp_kernel_common.h"

namespace lrp {

// ---- the tile kernel (RGB / RGBA / RGBAZ float) ----------------------------------
// Frames: the instantiation for batched launches whose wavefronts render their pixels for several consecutive frames of
// the batch (all frames share one geometry): the source coordinates of a wavefront's pixels are evaluated once, kept in
// registers, and every frame only requests its taps, interpolates and stores.
#ifndef LRP_TILE_MINWAVES_FRAMES
#define LRP_TILE_MINWAVES_FRAMES 4 // the frame-loop instantiations keep <= 128 VGPRs: they are bound by memory and need the wavefronts
#endif
// GeoRead: the instantiation whose pixels LOAD their source coordinates from a geometry-cache entry (lrp_geocache.h; the
// map is written as a side output by the plain path below when P.geo_mode == 1): nearest / bilinear, whole images; no lens
// math compiled in, the output lens is irrelevant (kRect by convention); one sample per pixel.  (num_samples 2-4 from an entry
// of sub-samples: lrp_ss_gather_kernel.h — a lane per sub-sample.)
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
  if constexpr ((OutLens == kEquirect || GeoRead) && InMode == kInRect) {
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
  // side output (nearest / bilinear, num_samples 2-4, P.geo_mode == 1): the entry of sub-samples of this geometry
  const bool geo_write_ss = !Frames && Interp != 2 && P.geo_mode == 1 && blockIdx.y == 0;

  for (int ssx = 0; ssx < ns; ++ssx) {
    const ColTerms col = column_terms<OutLens>(P, xe, ssx);
    for (int ssy = 0; ssy < ns; ++ssy) {
#pragma unroll
      for (int k = 0; k < kT2Rows; ++k) {
        const int yk = y_first + k;
        const int ye = yk < P.y_end ? yk : P.y_end - 1; // wave-uniform
        float sx, sy;
        pixel_source<OutLens, InMode>(P, col, ye, ssy, sx, sy);
        if constexpr (!Frames && Interp != 2)
          if (geo_write_ss) reinterpret_cast<vf2 *>(P.geo_xy)[geo_ss_map_index(xe, ye, P.out_w, ns * ns, ssx * ns + ssy)] = vf2{sx, sy};
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
  if (Interp != 2 && P.num_samples == 1 && P.batch_n > 1 && P.quad == 0 && P.geo_mode != 2) { // (the plain path: any rotation; the mirrored paths are bound by memory; GeoRead: a frame per workgroup)
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
      if (P.quad != 0 || P.num_samples != 1 || P.y_offset != 0 || P.y_end != P.out_h) return hipErrorInvalidValue;
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

} // namespace lrp
