// lrp_ss_gather_kernel.h — nearest / bilinear with num_samples 2-4 from an entry of sub-samples (lrp_geocache.h), a lane per
// SUB-SAMPLE.  The reference's loop (src/reproject.cpp:284-343) with its coordinates loaded instead of derived: the ns^2
// sub-samples of a pixel sit in ns^2 consecutive lanes in the reference's order — exactly the order of the entry
// (lrp_params.h geo_ss_map_index), so a wavefront's row of 64 / ns^2 pixels (16, 7, 4) is ONE coalesced 512-byte load —, every
// lane samples its own sub-sample (sample_pixels: the taps of the next rows in flight), the sub-samples of a pixel are summed in
// the reference's order by ss_ordered_sum (a DPP chain) and the pixel's last lane stores sum * (1 / ns^2) (+ the fused
// post_process).  The tile kernel (a lane per PIXEL) reads the same entry with 8 ns^2-byte strides between lanes: fine for
// ns 2 (two 16-byte loads per lane), 1.5-2.2x slower than COMPUTING the coordinates for ns 3, 4 — hence this kernel.  The
// entry is written by the first launch of the geometry, whichever sampler it is (tile kernel's computing path, or the window
// kernel's supersampling instantiations).  No lens math here: one instantiation per sampler, wrap mode and channel count.
#pragma once

#include "lrp_kernel_common.h"
#include "lrp_tile_kernel.h" // TileKernelFn

namespace lrp {

constexpr int kSsgRows = 4; // output rows per wavefront (a lane's sub-sample in four consecutive rows)

template <int Interp, bool Loop, int CH>
__global__ __launch_bounds__(kT2Threads, LRP_TILE_MINWAVES_FRAMES) void reproject_ss_gather_kernel(const KParams Pk) {
  static_assert(Interp == 0 || Interp == 1, "nearest / bilinear");
  KParams P = Pk; // src / dst: the frame being rendered
  if (Pk.batch_n > 0) {
    P.src = Pk.batch_src[blockIdx.y];
    P.dst = Pk.batch_dst[blockIdx.y];
  }
  int tx, ty;
  if (!xcd_tile(P.tiles_x, P.tiles_y, tx, ty)) return; // whole workgroup
  const int lane = (int)(threadIdx.x & 63u);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // this lane's pixel of the row and its sub-sample (the divisors are 4, 9, 16: a multiply and a shift)
  const int ns = P.num_samples, n = ns * ns, npx = 64 / n;
  const int l = min(lane, npx * n - 1); // (num_samples 3: lane 63 repeats lane 62 and stores nothing)
  const int p = n == 4 ? l >> 2 : n == 16 ? l >> 4 : (l * 57) >> 9;
  const int sub = l - p * n;
  const int x = tx * npx + p;
  const int xe = x < P.out_w ? x : P.out_w - 1; // lanes beyond the image repeat its last pixel and store nothing
  const bool owner = sub == n - 1 && lane < npx * n && x < P.out_w;
  const int y_first = (ty * kT2Waves + wave) * kSsgRows; // wave-uniform
  const SrcView src = source_view<Interp, CH>(P);
  const vf2 *const map = reinterpret_cast<const vf2 *>(P.geo_xy);
  auto coords = [&](int k, float &sx, float &sy) {
    const int yk = y_first + k;
    const vf2 v = __builtin_nontemporal_load(map + geo_ss_map_index(xe, yk < P.out_h ? yk : P.out_h - 1, P.out_w, n, sub));
    sx = v.x;
    sy = v.y;
  };
  auto finish = [&](int k, const Px<CH> &sample) {
    const Px<CH> a = ss_ordered_sum<CH>(sample, sub == 0, n);
    const int yk = y_first + k;
    if (owner && yk < P.out_h) store_px<CH, false>(P, (uint32_t)yk * (uint32_t)P.out_w + (uint32_t)x, a); // :338-341
  };
  sample_pixels<Interp, Loop, CH, kSsgRows>(P, src, coords, finish);
}

template <int Interp> struct SsGatherKernelTable {
  static TileKernelFn get(bool loop, int channels) {
    static const TileKernelFn table[2][3] = {
        {reproject_ss_gather_kernel<Interp, false, 3>, reproject_ss_gather_kernel<Interp, false, 4>, reproject_ss_gather_kernel<Interp, false, 5>},
        {reproject_ss_gather_kernel<Interp, true, 3>, reproject_ss_gather_kernel<Interp, true, 4>, reproject_ss_gather_kernel<Interp, true, 5>}};
    return table[loop ? 1 : 0][channels - 3];
  }
};

// P.geo_mode == 2 with an entry of sub-samples, P.num_samples 2-4, P.channels 3-5, whole images; interpolation nearest (0) or
// bilinear (1).
inline hipError_t launch_ss_gather_impl(KParams P, int interpolation, int in_mode, hipStream_t stream) {
  if (P.geo_mode != 2 || P.geo_xy == nullptr || P.num_samples < 2 || P.num_samples > 4 || P.channels < 3 || P.channels > 5 || P.y_offset != 0 ||
      P.y_end != P.out_h || (interpolation != 0 && interpolation != 1))
    return hipErrorInvalidValue;
  const int npx = 64 / (P.num_samples * P.num_samples);
  P.tiles_x = (P.out_w + npx - 1) / npx;
  P.tiles_y = (P.out_h + kSsgRows * kT2Waves - 1) / (kSsgRows * kT2Waves);
  if (P.tiles_x <= 0 || P.tiles_y <= 0) return hipSuccess;
  const bool loop = in_mode == kInEquirectLoop;
  const TileKernelFn fn = interpolation == 0 ? SsGatherKernelTable<0>::get(loop, P.channels) : SsGatherKernelTable<1>::get(loop, P.channels);
  const unsigned groups = P.batch_n > 0 ? (unsigned)P.batch_n : 1u;
  hipLaunchKernelGGL(fn, dim3((unsigned)(kXcds * xcd_rows(P.tiles_y) * P.tiles_x), groups), dim3(kT2Threads), 0, stream, P);
  return hipGetLastError();
}

} // namespace lrp
