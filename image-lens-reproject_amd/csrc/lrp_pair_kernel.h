// lrp_pair_kernel.h — alias pairs of in-view blocks rendered from ONE staged window by TWO wavefronts.
//
// A rectilinear view rendered into a full panorama (BASELINE configs[3]) appears twice: the reference has no hemisphere test
// (src/reproject.cpp:160-167 divides by -z whatever its sign), so the ray of panorama pixel (x + W/2, H-1-y) is the ray of (x, y)
// with x and z negated and lands on (nearly: different roundings) the same source texel.  The block (tx, ty) in front of the
// camera and its partner (tx + blocks_x / 2, blocks_y - 1 - ty) behind it read the same source window.  In the window kernel each
// of them is a wavefront of its own that stages that window for itself — at 20 KiB per wavefront eight of them fill a CU's LDS,
// two per SIMD, and the in-view part of the frame is bound by how many window round trips a CU has in flight and by what two
// wavefronts per SIMD can issue (profiles/r05_experiments_ab.txt item 2).  Here a WORKGROUP of two wavefronts renders the pair:
// one window per pass for both (the union of their extremes, half of its rows requested by either wavefront), every wavefront
// samples it with its own coordinates — half the window traffic, and sixteen (RGBAZ: twelve) wavefronts per CU instead of eight.
//
// Only pairs whose two blocks lie in view whole (every tap of every pixel unclamped and exact: the PAIR list of the
// geometry-cache entry, lrp_geo_lists.hip) come here; everything else stays with the window kernel.  Coordinates are loaded
// from the entry's map, taps are sampled by win_tier_raw (the window kernel's raw-tap tier): the same operations on the same
// operands, hence the same bits.  A pass whose union window does not fit the buffer gathers per pixel (sample_direct).
// Correctness never depends on the pairing: whatever two blocks share a workgroup, each is rendered from its own coordinates.
#pragma once

#include "lrp_kernel_common.h"
#include "lrp_win_plan.h"
#include "lrp_win_tiers.h"

namespace lrp {

#ifndef LRP_PAIR_CAP
#define LRP_PAIR_CAP 1270 // 16-byte slots of a pair's window: with the 128 bytes of the extremes just under 20 KiB, eight workgroups = sixteen wavefronts per CU
#endif
#ifndef LRP_PAIR_CAP5
#define LRP_PAIR_CAP5 1500 // RGBAZ (168 VGPRs: three wavefronts per SIMD, six workgroups per CU): 23.4 KiB + the two exchange buffers of its stores
#endif

template <int CH>
__global__ __launch_bounds__(128, CH == 5 ? 3 : 4) void reproject_pair_kernel(const KParams Pk) {
  static_assert(CH == 3 || CH == 4 || CH == 5, "RGB, RGBA or RGBAZ");
  constexpr int kCap = CH == 5 ? LRP_PAIR_CAP5 : LRP_PAIR_CAP;
  __shared__ float4 s_win[kCap];
  __shared__ int s_ext[2][4][4]; // [wavefront][pass]: lo_x, hi_x, lo_y, hi_y (float bits) of that wavefront's pixels of the pass
  const int lane = (int)(threadIdx.x & 63u);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // this workgroup's pair: the block in front of the camera; wavefront 1 renders its partner
  typedef const int32_t __attribute__((address_space(4))) *ScalarI;
  const uintptr_t list = reinterpret_cast<uintptr_t>(Pk.geo_pairs);
  const ScalarI e = reinterpret_cast<ScalarI>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(list >> 32)) << 32 |
                                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)list)) + (uint64_t)blockIdx.x * 8u);
  const int tx0 = e[0], ty0 = e[1];
  if (tx0 < 0) return; // the end of a shorter sub-list (both wavefronts)
  const int blocks_x = (int)geo_block_cols(Pk.out_w), blocks_y = (int)geo_image_block_rows(Pk.out_h);
  const int tx = wave == 0 ? tx0 : tx0 + (blocks_x >> 1), ty = wave == 0 ? ty0 : blocks_y - 1 - ty0;
  KParams P = Pk;
  const int frame = (int)blockIdx.y;
  P.src = Pk.batch_n > 0 ? Pk.batch_src[frame] : Pk.src;
  P.dst = Pk.batch_n > 0 ? Pk.batch_dst[frame] : Pk.dst;
  const SrcView src = source_view<2, CH>(P);
  float *out_lds = nullptr;
  if constexpr (CH == 5) {
    __shared__ __attribute__((aligned(16))) float s_out[2][320];
    out_lds = s_out[wave];
  }
  int prow, pcol;
  win_lane_pixel(lane, prow, pcol);
  const int x_blk = tx * kBlkW, y_blk = ty * kBlkH;
  const int xe = min(x_blk + pcol, P.out_w - 1);
  // the coordinates of this lane's four pixels (src/reproject.cpp:323-324, stored by the launch that filled the entry)
  float sx[4], sy[4];
  {
    typedef float vf2_ __attribute__((ext_vector_type(2)));
    const vf2_ *const map = reinterpret_cast<const vf2_ *>(P.geo_xy);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ye = min(y_blk + prow + kPassRows * k, P.out_h - 1);
      const vf2_ v = __builtin_nontemporal_load(map + geo_map_index(xe, ye, P.out_w));
      sx[k] = v.x;
      sy[k] = v.y;
    }
  }
  // The partner's pixel of (x, y) is (x + W/2, H-1-y): its block is this block upside down, its pass 3 - k reads what pass k
  // of the block in front of the camera reads.  Step i of the workgroup is pass i of wavefront 0 and pass 3 - i of wavefront 1.
  // the extremes of every step (the blocks are in view whole: the coordinates are >= 1, their bits order like integers)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = wave == 0 ? i : 3 - i;
    int lo_x = (int)f2u(sx[k]), hi_x = lo_x, lo_y = (int)f2u(sy[k]), hi_y = lo_y, d0 = 0, d1 = 0;
    wave_box(lo_x, hi_x, lo_y, hi_y, d0, d1);
    if (lane == 0) {
      const int k = i; // (stored by step)
      s_ext[wave][k][0] = lo_x;
      s_ext[wave][k][1] = hi_x;
      s_ext[wave][k][2] = lo_y;
      s_ext[wave][k][3] = hi_y;
    }
  }
  __syncthreads();
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)s_win;
  struct Win {
    int x_lo, y_lo, bw, bh, pitch;
    bool staged;
  };
  auto window_of = [&](int i, int n) { // the window of steps i .. i + n - 1 for both blocks (wave-uniform)
    Win w;
    int lo_x = min(s_ext[0][i][0], s_ext[1][i][0]), hi_x = max(s_ext[0][i][1], s_ext[1][i][1]);
    int lo_y = min(s_ext[0][i][2], s_ext[1][i][2]), hi_y = max(s_ext[0][i][3], s_ext[1][i][3]);
    if (n == 2) {
      lo_x = min(lo_x, min(s_ext[0][i + 1][0], s_ext[1][i + 1][0])), hi_x = max(hi_x, max(s_ext[0][i + 1][1], s_ext[1][i + 1][1]));
      lo_y = min(lo_y, min(s_ext[0][i + 1][2], s_ext[1][i + 1][2])), hi_y = max(hi_y, max(s_ext[0][i + 1][3], s_ext[1][i + 1][3]));
    }
    lo_x = __builtin_amdgcn_readfirstlane(lo_x), hi_x = __builtin_amdgcn_readfirstlane(hi_x);
    lo_y = __builtin_amdgcn_readfirstlane(lo_y), hi_y = __builtin_amdgcn_readfirstlane(hi_y);
    w.x_lo = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)lo_x)) - 1;
    w.y_lo = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)lo_y)) - 1;
    w.bw = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)hi_x)) + 2 - w.x_lo + 1;
    w.bh = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)hi_y)) + 2 - w.y_lo + 1;
    w.pitch = w.bw | 1;
    w.staged = w.bw <= 128 && win_slots_of_rows<CH>(w.pitch, w.bh) <= kCap; // (two DMA instructions per row at most)
    return w;
  };
#ifndef LRP_PAIR_HALVES
#define LRP_PAIR_HALVES 1 // steps 0-1 and 2-3 share ONE window where the union of both fits (one round trip and two barriers fewer)
#endif
  // a stage: one window and the `n` steps that sample it
  auto plan = [&](int i, Win &w, int &n) {
    n = 1;
    if (LRP_PAIR_HALVES != 0 && (i & 1) == 0) {
      w = window_of(i, 2);
      if (w.staged) {
        n = 2;
        return;
      }
    }
    w = window_of(i, 1);
  };
  // rows wave, wave + 2, ... of window `w` by this wavefront: LDS-DMA, one window row per instruction and 64 columns, lanes
  // beyond the width masked off (lrp_win_kernel.h issue())
  auto request = [&](const Win &w) {
    const int n_chunks = (w.bw + 63) >> 6;
    for (int chunk = 0; chunk < n_chunks; ++chunk)
      if (chunk * 64 + lane < w.bw) {
        const uint32_t lane_bytes = (uint32_t)(w.x_lo + chunk * 64 + lane) * (4u * CH);
        for (int r = wave; r < w.bh; r += 2) {
          const uint32_t lds = lds0 + (uint32_t)(r * w.pitch + chunk * 64) * 16u;
          const char *row = reinterpret_cast<const char *>(P.src) + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(w.y_lo + r) * src.row_bytes));
          if constexpr (CH == 3)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %2" : : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(lane_bytes), "s"(row) : "memory", "m0");
          else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(lane_bytes), "s"(row) : "memory", "m0");
          if constexpr (CH == 5) { // depth: the float plane behind the colour plane
            const uint32_t lds_d = lds0 + (uint32_t)(w.pitch * w.bh) * 16u + (uint32_t)(r * w.pitch + chunk * 64) * 4u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" : : "s"(__builtin_amdgcn_readfirstlane(lds_d)), "v"(lane_bytes), "s"(row + 16) : "memory", "m0");
          }
        }
      }
  };
#ifndef LRP_PAIR_PIPELINE
#define LRP_PAIR_PIPELINE 1 // the window of step i + 1 is requested behind the taps of step i, ahead of its arithmetic and its store (0: after the store)
#endif
  Win cur;
  int cur_n;
  plan(0, cur, cur_n);
  if (cur.staged) request(cur);
  int i = 0;
#pragma unroll 1
  while (i < 4) {
    const int next_i = i + cur_n;
    Win nxt = cur;
    int nxt_n = 1;
    if (next_i < 4) plan(next_i, nxt, nxt_n);
    const bool more = next_i < 4 && nxt.staged;
#pragma unroll 1
    for (int j = 0; j < cur_n; ++j) {
      const int step = i + j;
      const bool first = j == 0, last = j == cur_n - 1;
      const int k = wave == 0 ? step : 3 - step; // this wavefront's pass of the step
      // (k is wave-uniform but not a constant: the four coordinate pairs are selected, not indexed — no scratch)
      const float psx = k == 0 ? sx[0] : k == 1 ? sx[1] : k == 2 ? sx[2] : sx[3], psy = k == 0 ? sy[0] : k == 1 ? sy[1] : k == 2 ? sy[2] : sy[3];
      Rgba s;
      bool requested_next = false;
      if (cur.staged) {
        if (first) {
          // vmcnt retires in order: the window was requested BEFORE the previous step's store — at least one store per lane and
          // step —, so "at most one operation outstanding" means the window has landed (the first window: nothing younger)
          if (i == 0 || LRP_PAIR_PIPELINE == 0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
          // (a bare s_barrier: __syncthreads() is a fence as well — s_waitcnt vmcnt(0) lgkmcnt(0) in front of the barrier — and
          // would make every step wait for the previous step's STORE to be acknowledged; the waits that matter are the counted ones)
          asm volatile("s_barrier" ::: "memory"); // ... and so have the partner's rows
        }
        const float tx_ = __builtin_truncf(psx), ty_ = __builtin_truncf(psy);
        const int slot0 = __mul24((int)ty_ - 1 - cur.y_lo, cur.pitch) + ((int)tx_ - 1 - cur.x_lo);
        s = win_tier_raw<CH>(s_win + slot0, cur.pitch, reinterpret_cast<const float *>(s_win + cur.pitch * cur.bh) + slot0, psx - tx_, psy - ty_, [&]() {
          if (!last) return;
          // behind this wavefront's last read of the window: once both wavefronts are here the next window may overwrite it
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          asm volatile("s_barrier" ::: "memory");
          if (LRP_PAIR_PIPELINE != 0 && more) {
            request(nxt);
            requested_next = true;
          }
        });
      } else {
        // nothing of this step reads the window: the next one is requested in front of this step's gathers
        if (LRP_PAIR_PIPELINE != 0 && more) {
          request(nxt);
          requested_next = true;
        }
        if constexpr (CH == 5) {
          const Px<5> s5 = sample_direct<2, false, 5>(P, src, psx, psy);
          s = Rgba{s5.lo, s5.hi, s5.e};
        } else {
          s = sample_direct<2, false, 4, false, 4 * CH>(P, src, psx, psy);
        }
      }
      // src/reproject.cpp:334-341 with num_samples == 1: (0.0f + s) * 1.0f, the fused post_process, the store
      Rgba a4 = px_zero<4>();
      px_add<4>(a4, s);
      if constexpr (CH == 5) a4.e = 0.0f + s.e;
      const Px<CH> a{a4.lo, CH >= 4 ? a4.hi : f2{0.0f, 0.0f}, CH == 3 ? a4.hi.x : a4.e};
      const int y_top = y_blk + kPassRows * k;
      const int ye = min(y_top + prow, P.out_h - 1);
      bool stored_as_run = false;
      if constexpr (CH == 5) {
        if (x_blk + kBlkW <= P.out_w && y_top + kPassRows <= P.out_h) { // the pass lies in the image whole: four runs of 16 pixels
          float c[5];
          finish_px<5, true>(P, a, c);
          store_rgbaz_run<4>(P, out_lds, prow * kBlkW + pcol, (uint32_t)y_top * (uint32_t)P.out_w + (uint32_t)x_blk, P.out_w, c);
          stored_as_run = true;
        }
      }
      // (lanes / rows beyond the image hold the pixel they were clamped to and write its value to its place again)
      if (!stored_as_run) store_px<CH, true>(P, (uint32_t)ye * (uint32_t)P.out_w + (uint32_t)xe, a);
      if (last && more && !requested_next) request(nxt);
    }
    i = next_i;
    cur = nxt;
    cur_n = nxt_n;
  }
}

inline hipError_t launch_pair_bicubic(const KParams &P, hipStream_t stream) {
  if (P.geo_n_pairs == 0) return hipSuccess;
  if (P.geo_pairs == nullptr || P.geo_xy == nullptr || P.num_samples != 1) return hipErrorInvalidValue;
  const dim3 grid(P.geo_n_pairs, (unsigned)(P.batch_n > 0 ? P.batch_n : 1)), block(128);
  if (P.channels == 3)
    hipLaunchKernelGGL(reproject_pair_kernel<3>, grid, block, 0, stream, P);
  else if (P.channels == 4)
    hipLaunchKernelGGL(reproject_pair_kernel<4>, grid, block, 0, stream, P);
  else if (P.channels == 5)
    hipLaunchKernelGGL(reproject_pair_kernel<5>, grid, block, 0, stream, P);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

} // namespace lrp
