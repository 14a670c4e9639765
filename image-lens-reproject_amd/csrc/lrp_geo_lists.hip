// lrp_geo_lists.hip — rendering a geometry by block class (lrp_params.h "Block lists"):
//   geo_build_lists   once per geometry-cache entry, right behind the launch that wrote its class bytes: the WORK list
//                     (every block that is not a corner block, in XCD-interleaved launch order) and the corner RUNS;
//   corner_fill       the store-only kernel of the corner runs: no LDS window, no coordinates, no taps — every pixel of a
//                     run is the one value sample_bicubic gives when all 16 taps clamp to the same corner texel
//                     (src/reproject.cpp:109-148 with the clamped indices of :114-127 and weights of :130-131), finished
//                     like any other pixel (:334-341, fused post_process :421-437), written as whole contiguous row
//                     segments at 16 bytes per lane.
// In a rectilinear view rendered into a panorama (BASELINE configs[3]) 37 % of the blocks are corner blocks; inside the
// window kernel they occupy two-per-SIMD wave slots and leave as 320-byte pieces.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "lrp_corner_fill.h"

namespace lrp {

namespace {

constexpr int kListThreads = 1024;
constexpr int kListWaves = kListThreads / 64;

// Exclusive prefix sum of `v` over the workgroup, in thread order; *total = the workgroup's sum.
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t v, uint32_t *s_wave, uint32_t *total) {
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off);
    if (lane >= off) incl += up;
  }
  __syncthreads(); // (s_wave may still be read by the previous call)
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kListWaves; ++w) {
    const uint32_t t = s_wave[w];
    before += w < wave ? t : 0u;
    all += t;
  }
  *total = all;
  return before + incl - v;
}

// Workgroups 0-7: the work sub-list of XCD k.  The window kernel's launch deals workgroup w to XCD w % 8 and maps it to
// the block (j % blocks_x, (j / blocks_x) * 8 + w % 8), j = w / 8 (lrp_device.h xcd_tile, single rows of blocks); with
// alias pairs (lrp_win_kernel.h) consecutive workgroups of an XCD take the two blocks that read the same source texels in
// front of and behind the camera.  The sub-list is that sequence with the corner blocks removed, stored at 8 * position + k.
// Workgroup 8: the corner runs, rows of blocks top to bottom, groups of kGeoRunBlocks columns left to right.
__global__ __launch_bounds__(kListThreads) void geo_build_lists_kernel(const uint8_t *classes, uint32_t class_rows, const int32_t *box, int blocks_x,
                                                                       int blocks_y, int alias_pairs, uint32_t *header,
                                                                       int32_t *work, uint32_t *runs, int32_t *recs) {
  __shared__ uint32_t s_wave[kListWaves];
  const int k = (int)blockIdx.x;
  const bool alias = alias_pairs != 0 && (blocks_x & 1) == 0;
  if (k < kXcds) {
    const int rows_k = blocks_y > k ? (blocks_y - k + kXcds - 1) / kXcds : 0;
    const uint32_t n_items = (uint32_t)rows_k * (uint32_t)blocks_x;
    uint32_t base = 0;
    for (uint32_t i0 = 0; i0 < n_items; i0 += kListThreads) {
      const uint32_t j = i0 + threadIdx.x;
      int tx = 0, ty = 0;
      bool listed = false;
      if (j < n_items) {
        const int row = (int)(j / (uint32_t)blocks_x);
        tx = (int)(j - (uint32_t)row * (uint32_t)blocks_x);
        ty = row * kXcds + k;
        if (alias) {
          const bool rev = (tx & 1) != 0;
          tx = (tx >> 1) + (rev ? blocks_x >> 1 : 0);
          ty = rev ? blocks_y - 1 - ty : ty;
        }
        listed = classes[(size_t)tx * class_rows + (uint32_t)ty] == 0;
      }
      uint32_t total;
      const uint32_t pos = base + wg_exclusive_scan(listed ? 1u : 0u, s_wave, &total);
      if (listed) {
        work[2 * ((size_t)pos * kXcds + k)] = tx;
        work[2 * ((size_t)pos * kXcds + k) + 1] = ty;
        // ... and the block's box record beside it (lrp_params.h "recs")
        const int4 *const r = reinterpret_cast<const int4 *>(box + ((size_t)ty * (size_t)blocks_x + (size_t)tx) * 8);
        int4 *const o = reinterpret_cast<int4 *>(recs + 8 * ((size_t)pos * kXcds + k));
        o[0] = r[0];
        o[1] = r[1];
      }
      base += total;
    }
    if (threadIdx.x == 0) atomicMax(&header[0], base * kXcds); // entries of the interleaved list: 8 x the longest sub-list
    return;
  }
  const int groups = (blocks_x + kGeoRunBlocks - 1) / kGeoRunBlocks;
  const uint32_t n_items = (uint32_t)blocks_y * (uint32_t)groups;
  uint32_t base = 0, corner_blocks = 0;
  for (uint32_t i0 = 0; i0 < n_items; i0 += kListThreads) {
    const uint32_t item = i0 + threadIdx.x;
    const int row = (int)(item / (uint32_t)groups), c_first = (int)(item - (uint32_t)row * (uint32_t)groups) * kGeoRunBlocks;
    uint8_t cls[kGeoRunBlocks];
    uint32_t n_runs = 0;
#pragma unroll
    for (int c = 0; c < kGeoRunBlocks; ++c) {
      const int tx = c_first + c;
      cls[c] = (item < n_items && tx < blocks_x) ? classes[(size_t)tx * class_rows + (uint32_t)row] : (uint8_t)0;
      if (cls[c] != 0 && (c == 0 || cls[c] != cls[c - 1])) ++n_runs;
      corner_blocks += cls[c] != 0 ? 1u : 0u;
    }
    uint32_t total;
    uint32_t pos = base + wg_exclusive_scan(n_runs, s_wave, &total);
#pragma unroll
    for (int c = 0; c < kGeoRunBlocks; ++c) {
      if (cls[c] != 0 && (c == 0 || cls[c] != cls[c - 1])) {
        int len = 1;
        while (c + len < kGeoRunBlocks && cls[c + len] == cls[c]) ++len;
        runs[4 * (size_t)pos] = (uint32_t)row;
        runs[4 * (size_t)pos + 1] = (uint32_t)(c_first + c);
        runs[4 * (size_t)pos + 2] = (uint32_t)len;
        runs[4 * (size_t)pos + 3] = cls[c];
        ++pos;
      }
    }
    base += total;
  }
  uint32_t all_corner;
  (void)wg_exclusive_scan(corner_blocks, s_wave, &all_corner);
  if (threadIdx.x == 0) {
    header[1] = base;
    header[2] = all_corner;
    header[3] = (uint32_t)blocks_x * (uint32_t)blocks_y;
  }
}

// ---- the fill kernel -------------------------------------------------------------------------------------------------
// One workgroup = one run of one frame (blockIdx.y = frame of a batched launch): 16 pixel rows of up to
// kGeoRunBlocks * 16 pixels, four rows per wavefront (lrp_corner_fill.h).
constexpr int kFillWaves = 4;

template <int CH> __global__ __launch_bounds__(64 * kFillWaves) void corner_fill_kernel(const KParams P) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int frame = (int)blockIdx.y;
  const float *const src = P.batch_n > 0 ? P.batch_src[frame] : P.src;
  float *const dst = P.batch_n > 0 ? P.batch_dst[frame] : P.dst;
  const uint32_t first = blockIdx.x * 16u + (uint32_t)wave * (16u / kFillWaves);
  corner_fill_rows<CH>(P, src, dst, first, first + 16u / kFillWaves);
}

} // namespace

// ---- the census ------------------------------------------------------------------------------------------------------
// How many blocks of the geometry lie in the source whole (header[7]) and how many of THOSE have a window that the 10 KiB
// buffer of the four-wavefront kernels cannot stage, neither whole nor as two halves (header[6]: the plan of lrp_win_plan.h
// win_plan_block for RGB / RGBA, 64 columns, 640 slots) — such blocks fall to pass windows and per-pixel gathers there, and
// are what the big-window variant with its tap DMA is for (a cubemap's pole faces: three quarters of the blocks; its side
// faces: none).  The host reads the two counts with the list header and picks the variant per geometry.
constexpr int kCensusThreads = 256, kCensusCap = 640, kCensusCols = 64;
__global__ __launch_bounds__(kCensusThreads) void geo_census_kernel(const int32_t *box, uint32_t n_blocks, int in_w, int in_h, uint32_t *header) {
  const int one = (int)f2u(1.0f), x_hi = (int)f2u((float)(in_w - 2)), y_hi = (int)f2u((float)(in_h - 2));
  uint32_t in_view = 0, wide = 0;
  for (uint32_t i = blockIdx.x * kCensusThreads + threadIdx.x; i < n_blocks; i += gridDim.x * kCensusThreads) {
    const int32_t *const r = box + (size_t)i * 8;
    if ((r[6] & 7) != 7 || r[0] < one || r[1] >= x_hi || min(r[2], r[4]) < one || max(r[3], r[5]) >= y_hi) continue;
    ++in_view;
    const int x_first = (int)u2f((uint32_t)r[0]), x_last = (int)u2f((uint32_t)r[1]);
    const int ya_first = (int)u2f((uint32_t)r[2]), ya_last = (int)u2f((uint32_t)r[3]), yb_first = (int)u2f((uint32_t)r[4]), yb_last = (int)u2f((uint32_t)r[5]);
    const int bw = x_last - x_first + 4, pitch = bw | 1;
    const int bh = max(ya_last, yb_last) - min(ya_first, yb_first) + 4, a_rows = ya_last - ya_first + 4, b_rows = yb_last - yb_first + 4;
    const bool staged = bw <= kCensusCols && (pitch * bh <= kCensusCap || pitch * max(a_rows, b_rows) <= kCensusCap);
    wide += staged ? 0u : 1u;
  }
  // (wave-wide sums, one atomic per wavefront and count)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    in_view += __shfl_down(in_view, off);
    wide += __shfl_down(wide, off);
  }
  if ((threadIdx.x & 63u) == 0) {
    if (wide != 0) atomicAdd(&header[6], wide);
    if (in_view != 0) atomicAdd(&header[7], in_view);
  }
}

// Behind the launch that wrote the box records of an entry, on its stream (and behind launch_geo_build_lists where that
// runs: `clear_header` false).
hipError_t launch_geo_census(int32_t *box, int out_w, int out_h, int in_w, int in_h, bool clear_header, hipStream_t stream) {
  uint32_t *const header = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(box) + geo_lists_offset(out_w, out_h));
  if (clear_header) {
    const hipError_t e = hipMemsetAsync(header, 0, (size_t)kGeoListHeaderWords * 4, stream);
    if (e != hipSuccess) return e;
  }
  const uint32_t n_blocks = geo_block_cols(out_w) * geo_image_block_rows(out_h);
  const unsigned grid = (unsigned)std::min<uint32_t>(256u, (n_blocks + kCensusThreads - 1) / kCensusThreads);
  hipLaunchKernelGGL(geo_census_kernel, dim3(grid > 0 ? grid : 1), dim3(kCensusThreads), 0, stream, box, n_blocks, in_w, in_h, header);
  return hipGetLastError();
}

// Behind the launch that wrote the class bytes of an entry, on its stream.  `box` = the entry's first box record.
hipError_t launch_geo_build_lists(int32_t *box, int out_w, int out_h, int alias_pairs, hipStream_t stream) {
  uint8_t *const base = reinterpret_cast<uint8_t *>(box);
  uint32_t *const header = reinterpret_cast<uint32_t *>(base + geo_lists_offset(out_w, out_h));
  int32_t *const work = reinterpret_cast<int32_t *>(header + kGeoListHeaderWords);
  uint32_t *const runs = reinterpret_cast<uint32_t *>(work + 2 * geo_work_capacity(out_w, out_h));
  int32_t *const recs = reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(header) + geo_work_recs_offset(out_w, out_h));
  hipError_t e = hipMemsetAsync(header, 0, (size_t)kGeoListHeaderWords * 4, stream);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(work, 0xFF, geo_work_capacity(out_w, out_h) * 8, stream); // (-1, -1): nothing here
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(geo_build_lists_kernel, dim3(kXcds + 1), dim3(kListThreads), 0, stream, base + geo_class_offset(out_w, out_h),
                     geo_block_rows(out_h), box, (int)geo_block_cols(out_w), (int)geo_image_block_rows(out_h), alias_pairs, header, work, runs, recs);
  return hipGetLastError();
}

// The corner runs of P.geo_runs into P.dst (batched: into every P.batch_dst), P.channels in {3, 4, 5}, num_samples == 1.
hipError_t launch_corner_fill(const KParams &P, hipStream_t stream) {
  if (P.geo_n_runs == 0) return hipSuccess;
  if (P.geo_runs == nullptr || P.num_samples != 1) return hipErrorInvalidValue;
  const dim3 grid(P.geo_n_runs, (unsigned)(P.batch_n > 0 ? P.batch_n : 1)), block(64 * kFillWaves);
  if (P.channels == 3)
    hipLaunchKernelGGL(corner_fill_kernel<3>, grid, block, 0, stream, P);
  else if (P.channels == 4)
    hipLaunchKernelGGL(corner_fill_kernel<4>, grid, block, 0, stream, P);
  else if (P.channels == 5)
    hipLaunchKernelGGL(corner_fill_kernel<5>, grid, block, 0, stream, P);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

} // namespace lrp
