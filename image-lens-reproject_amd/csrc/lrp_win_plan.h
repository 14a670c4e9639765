// lrp_win_plan.h — the window kernel's block record and its window plan: from the wave-wide extremes of a block's source
// coordinates to the window origin / size, the tier (coefficient planes, raw taps, split halves, edge row / column, corner)
// and the LDS slots of everything the passes read.  Wave-uniform scalar code; lrp_win_kernel.h calls it once per block (and
// once per block and geometry when the extremes come from the geometry cache).
#pragma once

#include "lrp_kernel_common.h"

namespace lrp {

// ---- configuration of the window kernel (compile-time knobs; tools/ablate*.sh build variants) -----------------------
#ifndef LRP_WIN_CAP
#define LRP_WIN_CAP 640
#endif
#ifndef LRP_WIN_STRIP
#define LRP_WIN_STRIP 2
#endif
static_assert(LRP_WIN_STRIP <= kGeoStripRows, "geometry-cache entries hold block rows in multiples of kGeoStripRows (lrp_params.h)");
constexpr bool kWinCoef = true; // coefficient tier (below)
constexpr int kWinCap = LRP_WIN_CAP; // float4 texels per window buffer: 10 KiB per wavefront, 40 KiB per workgroup -> 4 workgroups / CU
#ifndef LRP_WIN_BLOCK_W
#define LRP_WIN_BLOCK_W 16
#endif
// Wavefronts per workgroup of the window kernel.  Its wavefronts share nothing (the window is
// wave-private), so a workgroup is ONE wavefront: each of the 16 wave slots of a CU is refilled
// the moment its wavefront retires instead of when the slowest of four does.
constexpr int kWinWaves = 1;
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
__device__ __forceinline__ void win_lane_pixel(int lane, int &prow, int &pcol) {
  prow = lane / kBlkW;
  pcol = lane & (kBlkW - 1);
}


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

// 16-byte LDS slots of `rows` window rows of `pitch` texels (RGBAZ: + the float plane of the depth channel behind the colour slots)
template <int CH> __device__ __forceinline__ int win_slots_of_rows(int pitch, int rows) {
  return CH == 5 ? pitch * rows + ((pitch * rows + 3) >> 2) : pitch * rows;
}
template <int CH, bool Split, bool Fat> __device__ __forceinline__ int win_raw_slots(const WinBlockT<Fat> &b) {
  return win_slots_of_rows<CH>(b.pitch, Split ? max(b.bh, b.bh2) : b.bh); // (bh2 == 0 unless split)
}
template <bool Fat> __device__ __forceinline__ void win_clear_block(WinBlockT<Fat> &b) {
  b.tier = 0;
  b.x_lo = b.y_lo = b.bw = b.bh = b.pitch = b.c_plane = b.c_base = b.tap_base = 0;
  b.y_lo2 = b.bh2 = 0;
  b.c_delta_stored[0] = b.c_delta_stored[1] = 0;
  b.iy0[0] = b.iy0[1] = b.iyn[0] = b.iyn[1] = 0;
}

// The thresholds of the plan as float bits in scalar registers (an int -> float conversion is a vector instruction: left to the
// compiler its result stays in a vector register for the whole kernel, and spills).
struct WinPlanLimits {
  int x_hi_bits, y_hi_bits;           // in_w - 2, in_h - 2: every tap of a pixel below them is unclamped
  int beyond_x_bits, beyond_y_bits;   // in_w + 1, in_h + 1: every tap of a pixel from there on clamps to the last column / row
};
__device__ __forceinline__ WinPlanLimits win_plan_limits(const KParams &P) {
  WinPlanLimits L;
  L.x_hi_bits = __builtin_amdgcn_readfirstlane((int)f2u((float)(P.in_w - 2)));
  L.y_hi_bits = __builtin_amdgcn_readfirstlane((int)f2u((float)(P.in_h - 2)));
  L.beyond_x_bits = __builtin_amdgcn_readfirstlane((int)f2u((float)(P.in_w + 1)));
  L.beyond_y_bits = __builtin_amdgcn_readfirstlane((int)f2u((float)(P.in_h + 1)));
  return L;
}

// Window of a block from the wave-wide extremes of its source coordinates (float bits): x range of the block, y ranges of
// its two halves (passes 0-1, 2-3).  exact_x / exact_y: int(s + k) == int(s) + k, k = -1 .. 2, holds for every pixel's x / y.
// CH: channels (RGBAZ windows carry a depth plane); Loop: the source wraps horizontally (no corner / edge blocks);
// Split / Edge: the instantiation stages split blocks / edge blocks; Cap: 16-byte slots of its window buffer.
template <int CH, bool Loop, bool Split, bool Edge, int Cap, int MaxCols = 64, bool Quad = false>
__device__ __forceinline__ void win_plan_block(WinBlockT<Quad> &block, const KParams &P, const WinPlanLimits &L, int w_lo_x, int w_hi_x, int w_lo_ya,
                                               int w_hi_ya, int w_lo_yb, int w_hi_yb, bool exact_x, bool exact_y) {
  constexpr int kPlanes = 3;
  // (planned on a local copy and written back whole: conditional stores through the reference get merged into stores through
  // a pointer phi before this function is inlined, and the caller's block record then stays in scratch memory)
  WinBlockT<Quad> b = block;
  const int w_lo_y = min(w_lo_ya, w_lo_yb), w_hi_y = max(w_hi_ya, w_hi_yb);
  const int one = (int)f2u(1.0f);
  // 1 <= s < extent - 2 for every pixel: every tap index is int(s) - 1 .. int(s) + 2, unclamped
  const bool in_x = exact_x && w_lo_x >= one && w_hi_x < L.x_hi_bits;
  const bool in_y = exact_y && w_lo_y >= one && w_hi_y < L.y_hi_bits;
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
    b.tier = (b.bw <= MaxCols && win_raw_slots<CH, Split>(b) <= Cap) ? 1 : 0; // (MaxCols: 64 texels per DMA instruction and window row)
    if (Split && P.win_split != 0 && b.tier == 0 && b.bw <= MaxCols) {
      const int a_lo = ya_first - 1, a_rows = ya_last + 2 - a_lo + 1, b_lo = yb_first - 1, b_rows = yb_last + 2 - b_lo + 1;
      if (win_slots_of_rows<CH>(b.pitch, max(a_rows, b_rows)) <= Cap) {
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
    if (!b.split() && win_raw_slots<CH, Split>(b) + kPlanes * b.pitch * (y_last - y_first + 1) <= Cap) b.tier |= 4;
    if (b.whole()) {
      b.iy0[0] = b.iy0[1] = y_first;
      b.iyn[0] = b.iyn[1] = y_last - y_first + 1;
    }
    b.c_plane = b.pitch * max(b.iyn[0], b.iyn[1]);
    if (kWinCoef && P.win_coef != 0 && b.staged() && !b.split() && win_raw_slots<CH, Split>(b) + kPlanes * b.c_plane <= Cap) b.tier |= 2;
    // planes behind the raw window plus, where there is room, one row and one column of slack:
    // the next block's (slightly different) window can then be requested while this block's
    // planes are still being read (see next_window)
    b.c_base = min(win_raw_slots<CH, Split>(b) + b.pitch + b.bh + 1, Cap - kPlanes * b.c_plane);
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
    const int sx_side = __builtin_amdgcn_readfirstlane(side(w_lo_x, w_hi_x, L.beyond_x_bits));
    const int sy_side = __builtin_amdgcn_readfirstlane(side(w_lo_y, w_hi_y, L.beyond_y_bits));
    if (sx_side != 0 && sy_side != 0) b.tier = (1 + (sx_side - 1) + 2 * (sy_side - 1)) << 3;
    // ... beyond one SIDE only (a rectilinear view inside a panorama: the rows above and below the view and the
    // columns left and right of it, another 37 % of the blocks): the same reasoning along one axis — all four tap
    // rows (columns) are the first or the last source row (column), the weight of that axis is 0 or 1 — and the
    // unclamped case along the other.  The block then reads ONE source row or column.
    else if (Edge && P.win_edge != 0 && sy_side != 0 && in_x) {
      const int x_first = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_lo_x));
      const int x_last = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_hi_x));
      b.x_lo = x_first - 1;
      b.bw = x_last + 2 - b.x_lo + 1;
      b.y_lo = sy_side == 2 ? P.in_h - 1 : 0;
      b.bh = 1;
      b.pitch = b.bw;
      b.c_base = win_raw_slots<CH, Split>(b); // the plane of vertical cubics, one per window texel (RGBAZ: + a float plane behind it)
      if (b.c_base + win_raw_slots<CH, Split>(b) <= Cap) b.tier = 1 | ((1 + (sy_side - 1)) << 6);
    } else if (Edge && P.win_edge != 0 && sx_side != 0 && in_y) {
      const int y_first = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_lo_y));
      const int y_last = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)w_hi_y));
      b.y_lo = y_first - 1;
      b.bh = y_last + 2 - b.y_lo + 1;
      b.x_lo = sx_side == 2 ? P.in_w - 1 : 0;
      b.bw = 1;
      b.pitch = 1;
      if (win_raw_slots<CH, Split>(b) <= Cap) b.tier = 1 | ((3 + (sx_side - 1)) << 6);
    }
  }
  block = b;
}

} // namespace lrp
