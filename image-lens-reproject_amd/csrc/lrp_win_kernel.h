// lrp_win_kernel.h — the bicubic LDS-window kernel and its launcher.
#pragma once

#include "lrp_kernel_common.h"
#include "lrp_corner_fill.h"
#include "lrp_tile_kernel.h" // TileKernelFn
#include "lrp_win_plan.h"
#include "lrp_win_tiers.h"

namespace lrp {

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
#if defined(LRP_WAVE_STAMPS) // diagnostic builds (tools/wave_timeline.py): start / end (100 MHz wall clock) and hardware slot of every wavefront of a launch
__device__ unsigned long long g_wave_stamps[3 * 65536];
#endif
#if defined(LRP_TIER_STATS) // diagnostic builds (tools/ablate.sh): blocks per tier (coefficients, raw taps, direct)
__device__ unsigned g_tier_stats[8]; // coefficient, raw, direct, corner, beyond a row, beyond a column, split
#endif


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
// SS: the supersampling instantiation, num_samples 2, 3 or 4 (the reference's --samples, src/reproject.cpp:294-298; its help text
// prescribes --scale 0.5 --samples 2, --scale 0.33334 --samples 3, --scale 0.25 --samples 4, src/main.cpp:192-196).  A LANE is one
// SUB-SAMPLE: the ns^2 sub-samples of a pixel sit in ns^2 consecutive lanes in the reference's order (sub = ns ssx + ssy), a pass
// is one output row of 64 / ns^2 pixels (16, 7 — lane 63 idles —, 4), a block four such rows: 256 samples whose taps span
// ~(npx ns + 3) x (4 ns + 3) source texels at the scales the option exists for — a window as compact as an ordinary block's, so
// the same plan, DMA and tiers serve it.  (Lanes as pixels and passes as sub-samples — round 5's layout, num_samples 2 only —
// would stage the footprint of 64 PIXELS per pass: 67 x 19 texels at --scale 0.25, beyond any buffer.)  The sub-samples of a
// pixel are summed in the reference's order (0.0f + s0 + s1 + ..., :334-336) by a chain of ns^2 - 1 DPP steps — step t: every
// lane adds its sample to the sum its left neighbour holds (wave_shr:1), so that after it the lane of sub-sample t holds
// 0 + s0 + ... + st — and the pixel's last lane stores sum * (1 / ns^2) (:338-341).  Plain blocks that compute their
// coordinates and, as a side output, write them into a geometry-cache entry of their own kind (one pair per sub-sample, the
// ns^2 of a pixel next to each other: a pass loads 512 contiguous bytes); GeoRead + SS: the instantiation that loads them — no
// lens math — and plans its windows from the loaded values.  No frame loop.
template <int OutLens, int InMode, int QMode, int CH, bool Frames = false, bool GeoRead = false, bool SS = false>
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
#define LRP_WIN_CAP_BIG 1120 // window slots of the big-window GeoRead variant (RGB / RGBA): 17.5 KiB per wavefront — nine wavefronts per CU (LDS is
                             // allocated in 512-byte granules: 1136 slots are eight again).  Round 4's 20 KiB held more whole and half windows at eight
                             // wavefronts; since the passes without a window fetch their taps (1027 slots) the ninth wavefront is worth more:
                             // rect -> equirect 4096^2 RGBA 179 -> 173 us single, 163 -> 157 batched; rect -> fisheye 146 -> 140 / 130 -> 124 (1060 / 1100 / 1120: level)
#endif
#ifndef LRP_WIN_CAP_BIG5
#define LRP_WIN_CAP_BIG5 1200 // ... RGBAZ: 18.75 KiB, so that with the 1.25 KiB exchange buffer of its stores eight wavefronts fit a CU's 160 KiB (1280 slots: seven; rect -> equirect RGBAZ + tonemap 330 -> 318 us single, 313 -> 300 batched)
#endif
#ifndef LRP_WIN_MINWAVES_BIG
#define LRP_WIN_MINWAVES_BIG 2
#endif
__global__ __launch_bounds__(kWinThreads, (GeoRead && OutLens == kEquirect) ? LRP_WIN_MINWAVES_BIG : CH == 5 ? LRP_WIN_MINWAVES5 : (QMode == 4 ? LRP_WIN_MINWAVES_RAYS : (Frames ? LRP_WIN_MINWAVES_FRAMES : (QMode >= 2 ? LRP_WIN_MINWAVES_AXIS : LRP_WIN_MINWAVES)))) void reproject_bicubic_win_kernel(const KParams Pk) {
  constexpr bool Quad = QMode != 0;
  constexpr bool MirX = QMode == 1 || QMode == 3 || QMode == 4, MirY = QMode == 1 || QMode == 2 || QMode == 4;
  constexpr bool kSharedRays = QMode == 4; // only the ray through the output lens is shared: per-image coordinates are stored like a plain block's
  static_assert(QMode != 4 || OutLens == kEquidistant, "shared rays: the equidistant target");
  constexpr int kAllMirrors = (MirX ? 1 : 0) | (MirY ? 2 : 0); // the image mirrored in every mirrored axis
  using WinBlock = WinBlockT<Quad>;
  static_assert(CH == 3 || CH == 4 || CH == 5, "window kernel: RGB, RGBA or RGBAZ");
  static_assert(!SS || (QMode == 0 && !Frames), "supersampling: plain blocks, no frame loop");
  constexpr int kBlockRows = SS ? 4 : kBlkH; // output rows of a block (SS: a pass is ONE output row)
  static_assert(QMode != 2 || (OutLens != kEquidistant && InMode != kInEquidistant), "rows-only mirroring goes through the column-separable source x");
  static_assert(QMode != 3 || OutLens == kRect, "columns-only mirroring needs vz == -1");
  static_assert(!GeoRead || (QMode == 0 && (OutLens == kRect || (OutLens == kEquirect && !Frames))),
                "GeoRead: plain blocks, one instantiation per source mode (+ the big-window variants)");
  // The big-window variant (GeoRead, "OutLens" kEquirect by convention; chosen by the host for a rectilinear view rendered
  // into a panorama, BASELINE configs[3]): the in-view blocks of that mapping are minified 3-5 x 1.5-3 — the window of a 16 x 4
  // PASS is ~67 x 11 texels, too wide for one DMA instruction per row and too large for 10 KiB next to three other
  // wavefronts' — so this variant holds 17.5 KiB (RGBAZ: 20 KiB) per wavefront (two wavefronts per SIMD, a third on one of them) and
  // stages the windows of single passes (and fetches the taps of the passes that have none: tap DMA below).  tools/microbench/row_gather.hip: rows of that shape arrive at 7.7 TB/s by
  // LDS-DMA with 8 wavefronts per CU, a window each in flight; per-pixel gathers of the same bytes at 4.5 TB/s in this kernel.
  constexpr bool kBigWin = GeoRead && OutLens == kEquirect;
  constexpr int kCap = kBigWin ? (CH == 5 ? LRP_WIN_CAP_BIG5 : LRP_WIN_CAP_BIG) : kWinCap; // 16-byte slots of this instantiation's window buffer
#ifndef LRP_BIG_PASSCOLS
#define LRP_BIG_PASSCOLS 64 // (128 — two DMA instructions per row for the wider ones — measured 1 % slower once such passes can fetch their taps instead)
#endif
  constexpr int kMaxPassCols = kBigWin ? LRP_BIG_PASSCOLS : 64; // widest pass window (texels): DMA instructions per window row = ceil(bw / 64)
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
  constexpr bool kEdge = InMode == kInRect;
  // Split blocks (WinBlockT::split) are compiled into the single-launch instantiations only: in the kernels with the frame
  // loop the extra code costs 2-7 % on mappings that have no such block (measured: equirect -> rect 91 -> 98 us, rect ->
  // equirect 212 -> 227), and what needs them — the 2048^2 faces of an 8192^2 panorama — arrives as single launches.
  // ... and for panorama sources only (a large panorama rendered into smaller views is where blocks are a little too large;
  // the rectilinear-source kernels lost 6 % to the extra code: rect -> equirect 210 -> 223 us).
  // (... and for the RGBAZ kernels of a rectilinear source: their per-pixel path is 20 gathers a pixel, and the pass windows
  // below pay there — rect -> equirect RGBAZ 300 -> 288 us, BASELINE configs[3] — while RGBA / RGB lose 4-6 %.)
#ifndef LRP_BIG_MAXCOLS
#define LRP_BIG_MAXCOLS 64 // widest block / half-block window of the big-window variant (pass windows: LRP_BIG_PASSCOLS)
#endif
  // (the big-window variant stages half blocks for every channel count: rect -> equirect 4096^2 RGBA 210 -> 203 us single, 191 -> 184 batched)
  constexpr bool kSplit = !Frames && (InMode == kInEquirect || InMode == kInEquirectLoop || (InMode == kInRect && CH == 5) || kBigWin);
  constexpr int kMaxStagedCols = kBigWin ? LRP_BIG_MAXCOLS : 64;
  // Pass windows (below) for rectilinear targets only — perspective views and cubemap faces out of a panorama; in the
  // fisheye-target kernels the extra code cost 2.5 % (equirect -> fisheye single launches 247 -> 253 us).
  constexpr bool kPassWin = (kSplit || (GeoRead && OutLens == kEquirect)) && (OutLens == kRect || GeoRead || (InMode == kInRect && CH == 5));
  // (the big-window variant has no coefficient tier — its blocks are minified, the planes rarely fit: rect -> equirect RGBAZ + tonemap
  // 258 -> 252.5 us batched, RGBA 158.4 -> 156.4)
  constexpr bool kCoefHere = !kBigWin;
  // Tap DMA (big-window variant; request_taps below) needs 1024 slots for the colour taps of a pass — and, RGBAZ, 1024 floats for their
  // depths behind them: exactly the 18.75 KiB window + the 1.25 KiB exchange buffer of the stores, which therefore lie in ONE
  // array (the exchange buffer is written behind the pixel's last tap read and read back before the next pass requests anything).
  constexpr bool kTapDma = kBigWin && kWinWaves == 1;
  constexpr int kOutSlots = (CH == 5 && kTapDma) ? 80 : 0; // 320 floats
  static_assert(!kTapDma || (CH == 5 ? (kCap + kOutSlots) * 16 >= 1024 * 20 : kCap >= 1024), "tap DMA: 16 taps x 64 pixels");
  // how the read-back of the taps avoids LDS bank conflicts (request_taps below): groups skewed by one slot each where the buffer
  // has the three slots to spare (conflict-free), else the quads rotated within their rows of 16 lanes (two-way)
  constexpr bool kTapRotate = CH == 5 || kCap < 1027;
  __shared__ float4 s_win[kWinWaves][kCap + kOutSlots];
#ifndef LRP_BIG_PASS_SLOTS
#define LRP_BIG_PASS_SLOTS 4096
#endif
  constexpr int kPassCap = (kTapDma && LRP_BIG_PASS_SLOTS < kCap) ? LRP_BIG_PASS_SLOTS : kCap; // largest window of a single pass (experiments: beyond the 1024 slots of its taps a window holds more texels than tap DMA fetches)

  // A listed launch may also hand every wavefront a share of the corner runs (Pk.geo_fill_per_wave row segments each,
  // lrp_corner_fill.h): a few store instructions at the very end of its life, when nothing else of it is live and nothing
  // waits for them — the stores of the out-of-view part of the frame then overlap the window round trips of the in-view
  // part inside ONE launch (as a kernel of its own in front of this one the fill costs its whole duration, beside it on a
  // second stream the fork / join costs more than it hides).
  // (compiled into the instantiations of the rectilinear source only — the source whose views leave most of a wider target
  // out of view; in the others, the headline's among them, the extra scalar state costs 7-10 spilled SGPRs for nothing)
  constexpr bool kListable = GeoRead && InMode == kInRect;
  auto fill_share = [&]() {
    if constexpr (kListable) {
      const uint32_t per_wave = Pk.geo_fill_per_wave;
      if (per_wave != 0 && blockIdx.x % Pk.geo_fill_stride == 0) {
        const uint32_t total = Pk.geo_n_runs * 16u, s0 = blockIdx.x / Pk.geo_fill_stride * per_wave;
        if (s0 < total)
          for (int f = 0; f < n_frames; ++f) corner_fill_rows<CH>(Pk, frame_src(f), frame_dst(f), s0, min(s0 + per_wave, total));
      }
    }
  };
  int tx, ty;
  // Listed launches (GeoRead, P.geo_work: lrp_params.h "Block lists"): workgroup i renders block geo_work[i] — the blocks of
  // the frame that are not corner blocks, already in XCD-interleaved order with alias pairs next to each other; the corner
  // blocks are written by the fill kernel (lrp_geo_lists.hip).  One block per wavefront.
  bool listed = false;
  // (the record beside the entry: in the big-window variant only — that is where listed launches run; in the four-wavefront
  // instantiations of the rectilinear source the seven scalars cost 17-29 spilled SGPRs and 14 spilled VGPRs)
  constexpr bool kListRec = kListable && kBigWin;
  int list_rec[7] = {0, 0, 0, 0, 0, 0, 0}; // a listed block's box record (scalar)
  bool have_list_rec = false;
  if constexpr (kListable) listed = Pk.geo_work != nullptr;
  if (listed) {
    typedef const int32_t __attribute__((address_space(4))) *ScalarI;
    const uintptr_t base = reinterpret_cast<uintptr_t>(Pk.geo_work);
    const uint64_t addr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32 |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base)) + (uint64_t)blockIdx.x * 8u;
    const ScalarI e = reinterpret_cast<ScalarI>(addr);
    tx = e[0];
    ty = e[1];
    // ... and the block's box record from beside the entry (lrp_params.h "recs"): both scalar loads are in flight together, the
    // window is planned one round trip earlier than from the box array (whose address depends on the entry)
    if (kListRec && Pk.geo_work_rec != nullptr) {
      const uintptr_t rbase = reinterpret_cast<uintptr_t>(Pk.geo_work_rec);
      const ScalarI r = reinterpret_cast<ScalarI>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rbase >> 32)) << 32 |
                                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rbase)) + (uint64_t)blockIdx.x * 32u);
#pragma unroll
      for (int i = 0; i < 7; ++i) list_rec[i] = r[i];
      have_list_rec = true;
    }
    if (tx < 0) { // the end of a shorter sub-list
      fill_share();
      return;
    }
  } else if (!xcd_tile<kWinXcdBand>(P.tiles_x, P.tiles_y, tx, ty, blockIdx.x)) return; // whole workgroup
#if defined(LRP_WAVE_STAMPS)
  const unsigned long long stamp_start = wall_clock64();
#endif
  // Alias pairs (mirrored strips of rectilinear -> equirectangular).  The reference has no
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
  constexpr bool kAliasPairs = (OutLens == kEquirect || GeoRead) && InMode == kInRect && kWinWaves == 1;
  int g_flip = 0;        // mirrored strips: the mirror image rendered by loop iteration g is g ^ g_flip
  bool g_reverse = false; // plain strips: iteration g renders block G-1-g
  if (kAliasPairs && P.alias_pairs != 0 && !listed) {
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
  const int G = GeoRead ? min(Gs, (P.y_end - P.y_offset + kBlockRows - 1) / kBlockRows - ty * Gs) : Gs;
  auto block_row = [&](int g) { return (kAliasPairs && g_reverse) ? G - 1 - g : g; }; // block of a plain strip rendered by iteration g
  // ... and its class byte (lrp_params.h)
  auto geo_classes = [&]() { return reinterpret_cast<uint8_t *>(P.geo_box) + geo_class_offset(P.out_w, P.out_h); };
  auto geo_class_index = [&](int g) { return (uint32_t)tx * geo_block_rows(P.out_h) + (uint32_t)(ty * Gs + block_row(g)); };
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
  // SS: this lane's pixel of the pass (pcol; every lane of a pass lies in one output row) and its sub-sample; wave-uniform:
  // sub-samples per pixel and pixels per pass.  (The divisors are 4, 9, 16 and 2, 3, 4: a multiply and a shift.)
  const int ss_ns = SS ? Pk.num_samples : 1, ss_n = ss_ns * ss_ns, ss_npx = SS ? 64 / ss_n : kBlkW;
  int ss_x = 0, ss_y = 0;
  bool ss_owner = false; // the lane that ends up with the pixel's sum: its last sub-sample
  if constexpr (SS) {
    const int l = min(lane, ss_npx * ss_n - 1); // (num_samples 3: lane 63 repeats lane 62 and stores nothing)
    pcol = ss_n == 4 ? l >> 2 : ss_n == 16 ? l >> 4 : (l * 57) >> 9;
    prow = 0;
    const int sub = l - pcol * ss_n;
    ss_x = ss_ns == 2 ? sub >> 1 : ss_ns == 4 ? sub >> 2 : (sub * 11) >> 5;
    ss_y = sub - ss_x * ss_ns;
    ss_owner = sub == ss_n - 1 && lane < ss_npx * ss_n;
  }
  const int x = tx * ((SS ? ss_npx : kBlkW) * kWinWaves) + wave * (SS ? ss_npx : kBlkW) + pcol;
  const int y_lane = P.y_offset + ty * (quad ? kBlkH : kBlockRows * Gs) + prow; // + kBlockRows * g + kPassRows * pass
  // output row of this lane's pixel in pass k of strip block g (plain blocks; SS: pass k is row k of the block)
  auto pixel_row = [&](int g, int k) { return y_lane + (quad ? 0 : kBlockRows * block_row(g)) + (SS ? k : kPassRows * k); };
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
  if constexpr (CH == 5 && kOutSlots != 0) {
    out_lds = reinterpret_cast<float *>(win0 + kCap);
  } else if constexpr (CH == 5) {
    __shared__ __attribute__((aligned(16))) float s_out[kWinWaves][320];
    out_lds = s_out[wave];
  }
  ColTerms col{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (!GeoRead) col = column_terms<OutLens>(P, xe, ss_x); // (SS: the terms of this lane's horizontal sub-sample)
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
  const WinPlanLimits plan_limits = win_plan_limits(P);
  auto slots_of_rows = [](int pitch, int rows) { return win_slots_of_rows<CH>(pitch, rows); };
  auto raw_slots = [&](const WinBlock &b) { return win_raw_slots<CH, kSplit>(b); };
  // (lrp_win_plan.h)
  auto plan_window = [&](WinBlock &b, int w_lo_x, int w_hi_x, int w_lo_ya, int w_hi_ya, int w_lo_yb, int w_hi_yb, bool exact_x, bool exact_y) {
    win_plan_block<CH, Loop, kSplit, kEdge, kCap, kMaxStagedCols>(b, P, plan_limits, w_lo_x, w_hi_x, w_lo_ya, w_hi_ya, w_lo_yb, w_hi_yb, exact_x, exact_y);
  };
  auto clear_block = [](WinBlock &b) { win_clear_block(b); };
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
  constexpr bool kStripPlan = Quad && !kSharedRays;
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

  // The window of a block whose coordinates b.sx / b.sy are known (computed by coords() below, or — GeoRead + SS — loaded):
  // per-pixel exactness, wave-wide extremes, the plan; a launch that fills a geometry-cache entry writes its side output here.
  auto plan_from_coords = [&](int g, WinBlock &b) {
    Extremes e;
#pragma unroll
    for (int k = 0; k < 4; ++k) note_pixel(e, k, b.sx[k], b.sy[k]);
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
            const int yk = pixel_row(g, k);
            const int ye = yk < qh ? yk : qh - 1;
            // (lanes / rows beyond the image hold the pixel they were clamped to and write its values to its place again)
            if constexpr (SS)
              map[geo_ss_map_index(xe, ye, P.out_w, ss_n, ss_x * ss_ns + ss_y)] = vf2{b.sx[k], b.sy[k]};
            else
              map[geo_map_index(xe, ye, P.out_w)] = vf2{b.sx[k], b.sy[k]};
          }
        }
        if constexpr (SS) return; // (an entry of sub-samples holds the map only: the reading launch plans from the loaded coordinates)
        const int words[7] = {planned ? e.lo_x : 0, planned ? e.hi_x : 0, planned ? e.lo_y[0] : 0, planned ? e.hi_y[0] : 0,
                              planned ? e.lo_y[1] : 0, planned ? e.hi_y[1] : 0,
                              (all_exact_x ? 1 : 0) | (all_exact_y ? 2 : 0) | (planned ? 4 : 0)};
        int bv = 0; // lane i = word i (written once per geometry: plain selects will do)
#pragma unroll
        for (int i = 0; i < 7; ++i) bv = lane == i ? words[i] : bv;
        if (lane < 8) P.geo_box[geo_block(g) * 8u + (uint32_t)lane] = bv;
        if (lane == 0) geo_classes()[geo_class_index(g)] = (uint8_t)(planned ? b.corner() : 0);
      }
    }
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
        const int yk = pixel_row(g, k);
        row_v[k] = row_term<OutLens>(P, yk < qh ? yk : qh - 1, ss_y);
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
    // (plain blocks: everything derived from the column terms alone — their products with the rotation matrix, the
    // column's share of the source lens — is loop-invariant, gets hoisted out of the block loop and then spilled to scratch
    // for the whole kernel: 80-100 MB of scratch traffic per 4K frame.  Opaque here, those few multiplies run per block.)
    ColTerms col_g = col;
    if constexpr (!Quad) asm volatile("" : "+v"(col_g.a), "+v"(col_g.b), "+v"(col_g.nx), "+v"(col_g.nz), "+v"(col_g.sx));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yk = pixel_row(g, k);
      const int ye = yk < qh ? yk : qh - 1;
      if (!quad)
        pixel_source_rt<OutLens, InMode>(P, col_g, row_v[k], ye, ss_y, b.sx[k], b.sy[k]);
      else
        quad_xy(gm, k, b.sx[k], b.sy[k]);
      // (shared rays: one pixel's rotation + source lens after the other — interleaved they do not fit the registers)
      if constexpr (kSharedRays) __builtin_amdgcn_sched_barrier(0);
    }
    plan_from_coords(g, b);
  };
  // GeoRead: the coordinates of block g and (geo_boxv, lane i = word i) its window extremes are requested by geo_fetch and
  // turned into a window plan by geo_plan, a few hundred instructions later
  int geo_boxv = 0;
  auto geo_fetch = [&](int g, WinBlock &b) {
    // (the extremes first: loads return in order, and the first block of a strip plans its window before anything else)
    if constexpr (!SS)
      if (!(kListRec && have_list_rec)) geo_boxv = __builtin_nontemporal_load(P.geo_box + (geo_block(g) * 8u + (uint32_t)(lane & 7)));
    const vf2 *const map = reinterpret_cast<const vf2 *>(P.geo_xy);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yk = pixel_row(g, k);
      const int ye = yk < qh ? yk : qh - 1;
      const vf2 v = __builtin_nontemporal_load(map + (SS ? geo_ss_map_index(xe, ye, P.out_w, ss_n, ss_x * ss_ns + ss_y) : geo_map_index(xe, ye, P.out_w)));
      b.sx[k] = v.x;
      b.sy[k] = v.y;
    }
  };
  // The classes of this strip's blocks (lrp_params.h geo_class_offset): one aligned word through the scalar cache when the
  // wavefront starts (strips of 1, 2 or 4 blocks: their class bytes lie in one word).  A corner block is then known before
  // anything is requested for it and skips geo_fetch: in a strip of corner blocks nothing stands between the stores
  // (a 4096^2 frame of corner blocks, every block waiting for its successor's record and coordinates: RGBA 100 -> 79 us,
  // RGBAZ + tonemap 167 -> 144 us).
  // (the big-window variant only: in every GeoRead instantiation of a source with corners the word costs 10 more spilled SGPRs at four
  // wavefronts per SIMD — rect -> fisheye single launches 164 -> 171 us)
  constexpr bool kGeoClasses = GeoRead && !Loop && kBigWin;
  uint32_t strip_classes = 0;
  if constexpr (kGeoClasses) {
    if (Gs <= 4 && (Gs & (Gs - 1)) == 0 && !listed) { // (a listed block is not a corner block)
      typedef const uint32_t __attribute__((address_space(4))) *ScalarU;
      const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)tx * geo_block_rows(P.out_h) + (uint32_t)(ty * Gs)));
      const uintptr_t base = reinterpret_cast<uintptr_t>(geo_classes());
      const uint64_t addr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32 |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base)) + (first & ~3u);
      strip_classes = *reinterpret_cast<ScalarU>(addr) >> (8u * (first & 3u));
    }
  }
  auto geo_class = [&](int g) -> int { // 0: plan from the box record; 1-4: corner block
    if constexpr (kGeoClasses) return (int)((strip_classes >> (8u * (uint32_t)block_row(g))) & 7u);
    return 0;
  };
  auto geo_plan = [&](WinBlock &b, int cls, int g_of_b) {
    clear_block(b);
    if constexpr (SS) { // (no records in an entry of sub-samples: the extremes come from the loaded coordinates)
      plan_from_coords(g_of_b, b);
      return;
    }
    if (cls != 0) {
      b.tier = cls << 3;
      return;
    }
    if constexpr (kListRec) {
      if (have_list_rec) { // (a listed launch renders one block per wavefront: this is its record)
        const int flags = list_rec[6];
        if ((flags & 4) != 0) plan_window(b, list_rec[0], list_rec[1], list_rec[2], list_rec[3], list_rec[4], list_rec[5], (flags & 1) != 0, (flags & 2) != 0);
        return;
      }
    }
    const int flags = __builtin_amdgcn_readlane(geo_boxv, 6);
    if ((flags & 4) != 0)
      plan_window(b, __builtin_amdgcn_readlane(geo_boxv, 0), __builtin_amdgcn_readlane(geo_boxv, 1),
                  __builtin_amdgcn_readlane(geo_boxv, 2), __builtin_amdgcn_readlane(geo_boxv, 3),
                  __builtin_amdgcn_readlane(geo_boxv, 4), __builtin_amdgcn_readlane(geo_boxv, 5), (flags & 1) != 0, (flags & 2) != 0);
  };
  // the one value of a corner block: sample_bicubic with all 16 taps on the corner texel (sample_direct's
  // one-column-and-one-row case, same operations)
  // (the texel through the scalar cache: its address is wave-uniform, a scalar load returns in ~200 cycles, occupies no vector
  // register until the arithmetic and stays out of the vmcnt order the window pipeline counts on — as a vector load it was a
  // second exposed round trip in front of every corner block's stores: a 4096^2 RGBA frame of corner blocks 109 -> 100 us)
  auto corner_value = [&](const WinBlock &b) {
    const int xh = (b.corner() - 1) & 1, yh = (b.corner() - 1) >> 1;
    const float fx = xh ? 1.0f : 0.0f, fy = yh ? 1.0f : 0.0f; // the clamped weights (src/reproject.cpp:130-131)
    const float hfx = 0.5f * fx, hfy = 0.5f * fy;
    const uint32_t off = (uint32_t)(yh ? P.in_h - 1 : 0) * src.row_bytes + (uint32_t)(xh ? in_w - 1 : 0) * (4u * CH);
    typedef const float __attribute__((address_space(4))) *ScalarF;
    const uintptr_t base = reinterpret_cast<uintptr_t>(P.src);
    const uint64_t addr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32 |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base)) + (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
    const ScalarF tp = reinterpret_cast<ScalarF>(addr);
    if constexpr (CH == 5) {
      const Px<5> t{f2{tp[0], tp[1]}, f2{tp[2], tp[3]}, tp[4]};
      const Px<5> k = cubic_px<5>(t, t, t, t, fy, hfy);
      const Px<5> r = cubic_px<5>(k, k, k, k, fx, hfx);
      return Rgba{r.lo, r.hi, r.e};
    } else {
      const Px<4> t{f2{tp[0], tp[1]}, f2{tp[2], CH == 3 ? 0.0f : tp[3]}, 0.0f}; // (RGB: the fourth component is unused)
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
        if constexpr (kBigWin) {
          // Per window row: M0, the request, the next row's LDS address and ONE vector add that moves the lanes' byte offsets to the
          // next source row — the scalar base is the frame's for every row.  Four rows per loop iteration: -1 % on the listed launches
          // of the big-window variant, nothing in the four-wavefront kernels (which keep the loop below).  (There: a 64-bit scalar
          // row pointer advanced by two adds, a counter, a compare and a branch per row — 8 scalar instructions per request, a fifth
          // of the kernel's.  The scalar add in place of the s_nop behind the write of M0 — a read-write scalar operand of the asm
          // statement — does not survive register allocation in the RGBAZ kernels: "illegal VGPR to SGPR copy".)
          const char *const base = reinterpret_cast<const char *>(frame);
          [[maybe_unused]] const char *const base_d = base + 16;
          uint32_t voff = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(kSplit ? b.first_row_of(half) : b.y_lo) * src.row_bytes)) + lane_bytes;
          uint32_t lds_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds);
          const uint32_t lds_step_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_step);
          [[maybe_unused]] uint32_t ldsd_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)win +
                                                                                       (uint32_t)(b.pitch * n_rows) * 16u + (uint32_t)(chunk * 64) * 4u));
          [[maybe_unused]] const uint32_t ldsd_step = (uint32_t)__builtin_amdgcn_readfirstlane(b.pitch * 4);
          auto request_row = [&]() {
            if constexpr (CH == 4)
              asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(__builtin_amdgcn_readfirstlane((int)lds_s)), "v"(voff), "s"(base) : "memory", "m0");
            else if constexpr (CH == 5) {
              // colour into the 16-byte slots of the row, depth into the row of the float plane behind the colour plane (the fifth
              // float's 16 bytes go into the scalar base)
              asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(__builtin_amdgcn_readfirstlane((int)lds_s)), "v"(voff), "s"(base) : "memory", "m0");
              asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" : : "s"(__builtin_amdgcn_readfirstlane((int)ldsd_s)), "v"(voff), "s"(base_d) : "memory", "m0");
              ldsd_s += ldsd_step;
            } else
              asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %2" : : "s"(__builtin_amdgcn_readfirstlane((int)lds_s)), "v"(voff), "s"(base) : "memory", "m0");
            lds_s += lds_step_s;
            voff += src.row_bytes;
          };
          int r = 0;
          for (; r + 4 <= n_rows; r += 4) {
            request_row();
            request_row();
            request_row();
            request_row();
          }
          for (; r < n_rows; ++r) request_row();
        } else {
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
    const int cls0 = geo_class(0);
    if (cls0 == 0) geo_fetch(0, cur);
    geo_plan(cur, cls0, 0);
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
    const bool planes_live = (kCoefHere && cur.coef()) || (kEdge && cur.edge() != 0 && cur.edge() < 3);
    dma_early = has_next() && (!planes_live || f_loop + 1 < n_frames || raw_slots(nxt) <= cur.c_base);
    if (dma_early) issue_next();
  };
  // The result of pass k of block g: num_samples == 1, (0.0f + s) * normalize (src/reproject.cpp:334-341), store.
  auto accumulate = [&](const Rgba &s) {
    Rgba a4 = px_zero<4>();
    px_add<4>(a4, s);
    if constexpr (CH == 5) a4.e = 0.0f + s.e;
    return Px<CH>{a4.lo, CH >= 4 ? a4.hi : f2{0.0f, 0.0f}, CH == 3 ? a4.hi.x : a4.e};
  };
  // where pass k of block g goes: (xo, yo) this lane's pixel; runs: the pass lies in the image whole (wave-uniform), `first` = its
  // first pixel, `row_step` = pixels from one of its four rows to the next, `mxo`: written right to left
  struct PassOut {
    int xo, yo, row_step;
    uint32_t first;
    bool whole, mxo;
  };
  auto pass_out = [&](int g, int k) {
    // Every lane stores: lanes / rows beyond the image have recomputed the pixel they were
    // clamped to (xe, ye) and write that same value to that same address again, so the
    // store is issued by every wavefront (the vmcnt(1) below counts on it).
    // (the four clamped rows of a mirrored strip are loop-invariant; hoisted they occupy four VGPRs for the whole
    // strip — which spilled — so the row is re-derived from an opaque copy here: an add and a min per pass)
    int y_base = y_lane;
    asm volatile("" : "+v"(y_base)); // (likewise not hoisted out of the frame loop)
    const int yk = y_base + (quad ? 0 : kBlockRows * block_row(g)) + (SS ? k : kPassRows * k);
    const int yc = yk < qh ? yk : qh - 1;
    const int gm = image_of(g);
    PassOut o;
    o.xo = (quad && (gm & 1)) ? P.out_w - 1 - xe : xe; // mirrored blocks write the mirrored pixel
    o.yo = (quad && (gm >> 1)) ? P.out_h - 1 - yc : yc;
    const int x_blk = tx * (kBlkW * kWinWaves) + wave * kBlkW;
    const int y_top = P.y_offset + ty * (quad ? kBlkH : kBlockRows * Gs) + (quad ? 0 : kBlockRows * block_row(g)) + (SS ? k : kPassRows * k);
    o.whole = x_blk + kBlkW <= qw && y_top + kPassRows <= qh;
    o.mxo = quad && (gm & 1);
    const bool myo = quad && (gm >> 1);
    o.first = (uint32_t)(myo ? P.out_h - 1 - y_top : y_top) * (uint32_t)P.out_w + (uint32_t)(o.mxo ? P.out_w - x_blk - kBlkW : x_blk);
    o.row_step = myo ? -P.out_w : P.out_w;
    return o;
  };
  // SS: does this lane's pixel of pass k of block g exist (its column inside the image, its row inside the band)?  Lanes beyond
  // recompute the pixel they were clamped to and store nothing.
  auto ss_inside = [&](int g, int k) { return x < qw && pixel_row(g, k) < qh; };
  auto emit = [&](int g, int k, const Rgba &s, auto as_runs, bool runs_rt = true) {
    if constexpr (SS) { // src/reproject.cpp:334-341: acc = 0.0f; acc += sample (ssx outer, ssy inner); dst = acc * normalize
      const Px<CH> sp{s.lo, CH >= 4 ? s.hi : f2{0.0f, 0.0f}, CH == 3 ? s.hi.x : s.e};
      const Px<CH> a = ss_ordered_sum<CH>(sp, (ss_x | ss_y) == 0, ss_n); // (lrp_kernel_common.h: the reference's order, a DPP chain)
      if (ss_owner && ss_inside(g, k)) {
        const PassOut o = pass_out(g, k);
        store_px<CH, false>(P, (uint32_t)o.yo * (uint32_t)P.out_w + (uint32_t)o.xo, a);
      }
      return;
    }
    const Px<CH> a = accumulate(s);
    const PassOut o = pass_out(g, k);
    if constexpr (CH == 5) {
     if constexpr (decltype(as_runs)::value) {
      // a pass that lies in the image whole (wave-uniform) leaves as four runs of 16 pixels (store_rgbaz_run).
      // Used where the stores are what a block costs: corner blocks (four stores and nothing else) in every kernel,
      // all blocks of a rectilinear view rendered into a panorama (kRunsEverywhere: most of that frame is out of
      // view or gathers minified taps; 383 -> 334 us).  In the VALU-bound kernels that interpolate from the LDS
      // window the exchange costs more than the stores gain (measured: 4-7 % slower).
      if (runs_rt && o.whole) {
        float c[5];
        finish_px<5, true>(P, a, c);
        store_rgbaz_run<4>(P, out_lds, prow * kBlkW + (o.mxo ? kBlkW - 1 - pcol : pcol), o.first, o.row_step, c);
        return;
      }
     }
    }
    store_px<CH, true>(P, (uint32_t)o.yo * (uint32_t)P.out_w + (uint32_t)o.xo, a);
  };
  // A corner block: every pixel is the one value `s`.  Finished (normalize, tonemap) once, stored four times; an RGBAZ pass that
  // lies in the image whole leaves as 80 sixteen-byte chunks of the repeating five-float pattern straight from registers (no
  // exchange through the LDS).  `before_last` runs in front of the last store (the next window's request).
  auto emit_corner = [&](int g, const Rgba &s, auto before_last) {
    Px<CH> a = accumulate(s);
    float c[5];
    if constexpr (SS) { // ns^2 equal sub-samples summed like any others (no lane needs another's), the pixels' last lanes store the four rows
#pragma unroll 1
      for (int t = 1; t < ss_n; ++t) px_add<CH>(a, Px<CH>{s.lo, CH >= 4 ? s.hi : f2{0.0f, 0.0f}, CH == 3 ? s.hi.x : s.e});
      finish_px<CH, false>(P, a, c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k == 3) before_last();
        if (ss_owner && ss_inside(g, k)) {
          const PassOut o = pass_out(g, k);
          store_texel_nt<CH>(P.dst + (size_t)((uint32_t)o.yo * (uint32_t)P.out_w + (uint32_t)o.xo) * CH, c);
        }
      }
      return;
    }
    finish_px<CH, true>(P, a, c);
    typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
    v4f_a4 q0{0.0f, 0.0f, 0.0f, 0.0f}, q1{0.0f, 0.0f, 0.0f, 0.0f};
    int chunk_row = 0, chunk_col = 0;
    if constexpr (CH == 5) {
      // chunk `lane` = chunk cc of run row r (20 chunks per row of 16 pixels); chunks 64-79 (lanes 0-15) = row 3, chunks 4-19
      chunk_row = (lane * 3277) >> 16;
      chunk_col = lane - 20 * chunk_row;
      auto pattern = [&](int chunk) { // floats 4 chunk .. 4 chunk + 3 of c[0] c[1] c[2] c[3] c[4] c[0] ...
        const int m = (4 * chunk) % 5;
        auto at = [&](int i) { const int j = (m + i) % 5; return j == 0 ? c[0] : j == 1 ? c[1] : j == 2 ? c[2] : j == 3 ? c[3] : c[4]; };
        return v4f_a4{at(0), at(1), at(2), at(3)};
      };
      q0 = pattern(chunk_col);
      q1 = pattern(4 + lane);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k == 3) before_last();
      const PassOut o = pass_out(g, k);
      if constexpr (CH == 5) {
        if (o.whole) { // (a constant run reads the same mirrored or not)
          float *const row0 = P.dst + (size_t)o.first * 5;
          const ptrdiff_t step = (ptrdiff_t)o.row_step * 5;
          __builtin_nontemporal_store(q0, reinterpret_cast<v4f_a4 *>(row0 + chunk_row * step + 4 * chunk_col));
          if (lane < 16) __builtin_nontemporal_store(q1, reinterpret_cast<v4f_a4 *>(row0 + 3 * step + 16 + 4 * lane));
          continue;
        }
      }
      store_texel_nt<CH>(P.dst + (size_t)((uint32_t)o.yo * (uint32_t)P.out_w + (uint32_t)o.xo) * CH, c);
    }
  };
  // Pass windows (kernels with split blocks): a block whose two half windows do not fit either (a pole face of a
  // cubemap: the panorama's rows converge) still has passes — 16 x 4 pixels — whose own window fits.  Planned per pass
  // from the wave-wide extremes of that pass's coordinates, fetched, waited for and read on the spot; the other
  // wavefronts of the SIMD cover the round trip (requesting the window of pass k + 1 behind the taps of pass k, like the
  // second half of a split block, measured slower: 120 against 112 us per pole face, 14 spilled registers).
  // False: this pass gathers per pixel.
  // one pixel from a staged pass window `w`: the raw-tap tier (lrp_win_tiers.h)
  auto window_sample = [&](const WinBlock &w, float psx, float psy, bool last_pass) -> Rgba {
    const float tx_ = __builtin_truncf(psx), ty_ = __builtin_truncf(psy);
    const int slot0 = __mul24((int)ty_ - 1 - w.y_lo, w.pitch) + ((int)tx_ - 1 - w.x_lo);
    return win_tier_raw<CH>(win0 + slot0, w.pitch, reinterpret_cast<const float *>(win0 + w.pitch * w.bh) + slot0, psx - tx_, psy - ty_, [&]() {
      if (last_pass) next_window(); // behind the last reads of this pass's window
    });
  };
  // window of one pass (or of two passes that read the same source rows) from the wave-wide extremes of its coordinates;
  // false: too wide or too large for the buffer
  // A pass spans more than kMaxPassCols texels for certain when two of its pixels lie that far apart: the first and the last
  // pixel of its first row, two v_readlane instead of the wave-wide reductions (a quarter of the in-view passes of BASELINE
  // configs[3] end here and fetch their taps).  With a < b: int(b) - int(a) > b - a - 1, so b - a >= kMaxPassCols - 3 means a
  // window of int(b) - int(a) + 4 > kMaxPassCols columns.  (NaN compares false: the full plan decides.)
  auto pass_too_wide = [&](float psx) -> bool {
    const float xa = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(psx), 0));
    const float xb = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(psx), kBlkW - 1));
    return __builtin_fabsf(xb - xa) >= (float)(kMaxPassCols - 3);
  };
  auto plan_pass_window = [&](WinBlock &w, int lo_x, int hi_x, int lo_y, int hi_y) -> bool {
    if (kBigWin && pass_too_wide(u2f((uint32_t)lo_x))) return false;
    int d0 = 0, d1 = 0;
    wave_box(lo_x, hi_x, lo_y, hi_y, d0, d1); // (interior: the coordinates are >= 1, their bits order like integers)
    clear_block(w);
    w.x_lo = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)lo_x)) - 1;
    w.y_lo = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)lo_y)) - 1;
    w.bw = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)hi_x)) + 2 - w.x_lo + 1;
    w.bh = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)hi_y)) + 2 - w.y_lo + 1;
    w.pitch = w.bw | 1;
    if (w.bw > kMaxPassCols || slots_of_rows(w.pitch, w.bh) > kPassCap) return false;
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
  // Tap DMA (big-window variant).  A pass whose window fits no buffer — the view minified four times and more: the 4 x 4
  // footprints of neighbouring pixels no longer overlap, a window would stage mostly texels nobody reads — gathers 16 taps per
  // pixel; a gather per lane touches 64 different cache lines per instruction (BASELINE configs[3]: 38 % of the in-view pixels,
  // 45 M of the launch's 54 M L1 accesses, PMC).  Here the FOUR LANES OF A QUAD fetch the four consecutive texels of one pixel's
  // tap row — 64 bytes, one line or two — by LDS-DMA: instruction (t, r) fetches row r of pixel t of every quad into 64
  // consecutive slots, 16 instructions (RGBAZ: + 16 for the depths) land the pass's 1024 taps in LDS at 16 slots per pixel, and
  // the pixel reads them back as a window of pitch 64 (the raw-tap tier: same taps, same operations, same bits).  Bank
  // conflicts: RGBA / RGB skew the four groups by one slot each (conflict-free); RGBAZ has not one spare byte, its quads are
  // rotated by t positions within their row of 16 lanes instead (two-way).
  // The passes of a block with nothing staged (the rolled loop below): pass k + 1 is planned and REQUESTED behind the last tap
  // read of pass k — its own window where that fits, else its taps — so that its round trip runs under the arithmetic and the
  // store of pass k (a wavefront of this variant has one neighbour on its SIMD to cover for it, not three).
  struct PassWin {
    int x_lo, y_lo, bw, bh, pitch;
  };
  const uint32_t pass_lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)win0);
  auto plan_pass = [&](PassWin &w, float psx, float psy) -> bool { // (interior: the coordinates are >= 1, their bits order like integers)
    if (pass_too_wide(psx)) return false;
    int lo_x = (int)f2u(psx), hi_x = lo_x, lo_y = (int)f2u(psy), hi_y = lo_y, d0 = 0, d1 = 0;
    wave_box(lo_x, hi_x, lo_y, hi_y, d0, d1);
    w.x_lo = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)lo_x)) - 1;
    w.y_lo = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)lo_y)) - 1;
    w.bw = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)hi_x)) + 2 - w.x_lo + 1;
    w.bh = __builtin_amdgcn_readfirstlane((int)u2f((uint32_t)hi_y)) + 2 - w.y_lo + 1;
    w.pitch = w.bw | 1;
    return w.bw <= kMaxPassCols && slots_of_rows(w.pitch, w.bh) <= kPassCap;
  };
  auto request_pass = [&](const PassWin &w) { // one window row and 64 columns per instruction, lanes beyond the width masked off (issue())
    const int n_chunks = (w.bw + 63) >> 6;
    for (int chunk = 0; chunk < n_chunks; ++chunk)
      if (chunk * 64 + lane < w.bw) {
        const uint32_t lane_bytes = (uint32_t)(w.x_lo + chunk * 64 + lane) * (4u * CH);
        const char *row = reinterpret_cast<const char *>(P.src) + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)w.y_lo * src.row_bytes));
        uint32_t lds = pass_lds0 + (uint32_t)(chunk * 64) * 16u, lds_d = pass_lds0 + (uint32_t)(w.pitch * w.bh) * 16u + (uint32_t)(chunk * 64) * 4u;
        for (int r = 0; r < w.bh; ++r) {
          if constexpr (CH == 3)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %2" : : "s"(lds), "v"(lane_bytes), "s"(row) : "memory", "m0");
          else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds), "v"(lane_bytes), "s"(row) : "memory", "m0");
          if constexpr (CH == 5) // depth: the float plane behind the colour plane
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" : : "s"(lds_d), "v"(lane_bytes), "s"(row + 16) : "memory", "m0");
          row += src.row_bytes;
          lds += (uint32_t)w.pitch * 16u;
          lds_d += (uint32_t)w.pitch * 4u;
        }
      }
  };
  // (RGBAZ: the depths land behind the 1024 colour slots)
  auto request_taps = [&](float psx, float psy) {
    constexpr uint32_t depth_at = 1024u * 16u;
    const float tx_ = __builtin_truncf(psx), ty_ = __builtin_truncf(psy);
    constexpr uint32_t T = 4u * CH;
    const uint32_t v0 = __umul24((uint32_t)((int)ty_ - 1), src.row_bytes) + (uint32_t)((int)tx_ - 1) * T; // tap (0, 0) of this lane's pixel
    const char *const base = reinterpret_cast<const char *>(P.src);
    const int j = lane & 3;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // the offset of the pixel this lane helps to fetch in group t: pixel t of this lane's quad (quad_perm broadcast), RGBAZ: of
      // the quad t positions further down its row of 16 lanes (row_ror: lane i takes lane i - 4 t)
      int got;
      if (t == 0) got = __builtin_amdgcn_mov_dpp((int)v0, 0x00, 0xF, 0xF, false);
      else if (t == 1) got = __builtin_amdgcn_mov_dpp((int)v0, 0x55, 0xF, 0xF, false);
      else if (t == 2) got = __builtin_amdgcn_mov_dpp((int)v0, 0xAA, 0xF, 0xF, false);
      else got = __builtin_amdgcn_mov_dpp((int)v0, 0xFF, 0xF, 0xF, false);
      if constexpr (kTapRotate) {
        if (t == 1) got = __builtin_amdgcn_mov_dpp(got, 0x120 + 4, 0xF, 0xF, false);
        else if (t == 2) got = __builtin_amdgcn_mov_dpp(got, 0x120 + 8, 0xF, 0xF, false);
        else if (t == 3) got = __builtin_amdgcn_mov_dpp(got, 0x120 + 12, 0xF, 0xF, false);
      }
      const uint32_t vt = (uint32_t)got + (uint32_t)j * T;
      // the four tap rows of the group with ONE write of M0: the instruction offset moves the LDS address AND the global one, so
      // row r (LDS: 1024 r bytes further, depth plane: 256 r) takes a scalar base that many bytes lower
      {
        const uint32_t o0 = vt, o1 = vt + src.row_bytes, o2 = vt + 2u * src.row_bytes, o3 = vt + 3u * src.row_bytes;
        const uint32_t lds = pass_lds0 + (uint32_t)((4 * t * 64 + (kTapRotate ? 0 : t)) * 16);
        if constexpr (CH == 3)
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %5\n\tglobal_load_lds_dwordx3 %2, %6 offset:1024\n\t"
                       "global_load_lds_dwordx3 %3, %7 offset:2048\n\tglobal_load_lds_dwordx3 %4, %8 offset:3072"
                       :
                       : "s"(lds), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base), "s"(base - 1024), "s"(base - 2048), "s"(base - 3072)
                       : "memory", "m0");
        else
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\tglobal_load_lds_dwordx4 %2, %6 offset:1024\n\t"
                       "global_load_lds_dwordx4 %3, %7 offset:2048\n\tglobal_load_lds_dwordx4 %4, %8 offset:3072"
                       :
                       : "s"(lds), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base), "s"(base - 1024), "s"(base - 2048), "s"(base - 3072)
                       : "memory", "m0");
        if constexpr (CH == 5) {
          const uint32_t lds_d = pass_lds0 + depth_at + (uint32_t)(4 * t * 64 * 4);
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %5\n\tglobal_load_lds_dword %2, %6 offset:256\n\t"
                       "global_load_lds_dword %3, %7 offset:512\n\tglobal_load_lds_dword %4, %8 offset:768"
                       :
                       : "s"(lds_d), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base + 16), "s"(base + 16 - 256), "s"(base + 16 - 512), "s"(base + 16 - 768)
                       : "memory", "m0");
        }
      }
    }
  };
  // plans pass `k` (coordinates psx, psy) and requests what it will read: 1 its own window `w`, 2 its taps, 0 nothing (it gathers)
  auto prepare_pass = [&](PassWin &w, float psx, float psy) -> int {
    if (!all_interior(psx, psy, 1.0f, src.x_hi, src.y_hi, 2.0f)) return 0;
    const bool fits = kPassWin && P.win_split != 0 && plan_pass(w, psx, psy);
    if (!fits && !(kTapDma && P.win_tapdma != 0)) return 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // every earlier read of the buffer (and of the exchange buffer in its tail) has returned
    if (fits) {
      request_pass(w);
      return 1;
    }
    request_taps(psx, psy);
    return 2;
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
    // (a launch that writes the geometry cache has the stores of coords(g + 1) in flight as well: it waits for everything)
    // (SS: the stores are predicated — a pass whose pixels all lie beyond the image issues none — so nothing may be assumed
    // to be younger than the window's request)
    if ((g == 0 && f == 0) || !dma_early || geo_write || SS)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the window was the last thing requested
    else
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); // window g has landed; block g-1's last store may be in flight
    if constexpr (GeoRead)
      if (g + 1 < G && f == 0 && geo_class(g + 1) == 0) geo_fetch(g + 1, nxt);
    const float4 *const win = win0;
    // The tier of this block in a scalar register for the branches below: carried through the block loop inside `cur` it
    // ends up in a VGPR (the kernel is at the SGPR limit), and every test of it then costs a v_and + v_cmp and the
    // branch condition is re-materialised through v_cndmask / v_cmp at each use — seven VALU instructions per pass.
    const int tier = __builtin_amdgcn_readfirstlane(cur.tier);
    const bool t_coef = kCoefHere && (tier & 2) != 0, t_staged = (tier & 1) != 0, t_whole = (tier & 4) != 0;
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
        if (g + 1 < G && last_frame) geo_plan(nxt, geo_class(g + 1), g + 1);
      emit_corner(g, cs, [&]() { next_window(); });
      if (!dma_early && has_next()) issue_next();
      if (last_frame) cur = nxt;
      continue;
    }
    // The big-window variant: a block with nothing staged — its four passes take a window of their own, tap DMA or gathers,
    // decided pass by pass — runs them in ONE rolled loop: the three paths exist once instead of four times (the unrolled passes
    // below then hold the staged tiers only), 27 KB less code and the registers of the staged tiers are not shared with the
    // gathers' 80 tap registers.
    constexpr bool kRolledUnstaged = kBigWin;
    if constexpr (kRolledUnstaged) {
      if (!t_staged && !t_coef && t_edge == 0) {
        // (k is wave-uniform but not a constant: the coordinates are selected, not indexed — no scratch)
        // (... and from opaque copies: a chain of selects over the elements of one array is turned back into an indexed load)
        float x0 = cur.sx[0], x1 = cur.sx[1], x2 = cur.sx[2], x3 = cur.sx[3], y0 = cur.sy[0], y1 = cur.sy[1], y2 = cur.sy[2], y3 = cur.sy[3];
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
        auto pass_x = [&](int k) { return k == 0 ? x0 : k == 1 ? x1 : k == 2 ? x2 : x3; };
        auto pass_y = [&](int k) { return k == 0 ? y0 : k == 1 ? y1 : k == 2 ? y2 : y3; };
        // (RGBAZ: on the spot — its taps fill the buffer to the last byte, the exchange buffer of the stores in its tail included,
        // and can only be requested once the previous pass has left; with the request in two places its kernel is 3 % slower)
        constexpr bool kPipeline = kOutSlots == 0;
        PassWin w;
        int kind = 0;
        bool behind_store = true; // what this pass reads was requested behind the previous pass's store: nothing younger in flight
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
          if (k == 3 && g + 1 < G && last_frame) geo_plan(nxt, geo_class(g + 1), g + 1);
          const float psx = pass_x(k), psy = pass_y(k);
          if constexpr (!kPipeline) { // planned, requested, waited for and read on the spot
            const bool last_pass = k == 3;
            Rgba s;
            if (kPassWin && P.win_split != 0 && pass_window(psx, psy, s, last_pass)) {
              // (rendered from the window of this pass)
            } else if (kTapDma && P.win_tapdma != 0 && all_interior(psx, psy, 1.0f, src.x_hi, src.y_hi, 2.0f)) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // every earlier read of the buffer (and of the exchange buffer in its tail) has returned
              request_taps(psx, psy);
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the taps (and every older store)
              const int j = lane & 3;
              const int slot0 = kTapRotate ? (256 * j + (lane & 48) + ((((lane >> 2) + j) & 3) << 2)) : (257 * j + (lane & ~3));
              s = win_tier_raw<CH, true>(win0 + slot0, 64, reinterpret_cast<const float *>(win0 + 1024) + slot0, psx - __builtin_truncf(psx), psy - __builtin_truncf(psy), [&]() {
                if (last_pass) next_window(); // behind the last reads of the taps
              });
            } else {
              if (last_pass) next_window(); // nothing staged: no tap of this block reads the window
              if constexpr (CH == 5) {
                const Px<5> s5 = sample_direct<2, Loop, 5, LRP_WIN_MINWAVES5 >= 4>(P, src, psx, psy);
                s = Rgba{s5.lo, s5.hi, s5.e};
              } else {
                s = sample_direct<2, Loop, 4, (LRP_WIN_MINWAVES >= 5), 4 * CH>(P, src, psx, psy);
              }
            }
            emit(g, k, s, std::integral_constant<bool, kRunsEverywhere>{}, !GeoRead || P.rgbaz_runs != 0);
            continue;
          }
          if (k == 0) {
            kind = prepare_pass(w, psx, psy);
            behind_store = true;
          }
          const float tx_ = __builtin_truncf(psx), ty_ = __builtin_truncf(psy);
          int next_kind = 0;
          PassWin wn = w;
          // behind the pass's last read of the buffer: the next block's window (last pass), else what pass k + 1 reads
          auto after_reads = [&]() {
            if (k == 3)
              next_window();
            else if (kPipeline)
              next_kind = prepare_pass(wn, pass_x(k + 1), pass_y(k + 1));
          };
          Rgba s;
          if (kind != 0) {
            // vmcnt retires in order: what this pass reads was requested in front of the previous pass's store, or behind it
            if (behind_store)
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else
              asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            if (kind == 1) {
              const int slot0 = __mul24((int)ty_ - 1 - w.y_lo, w.pitch) + ((int)tx_ - 1 - w.x_lo);
              s = win_tier_raw<CH>(win0 + slot0, w.pitch, reinterpret_cast<const float *>(win0 + w.pitch * w.bh) + slot0, psx - tx_, psy - ty_, after_reads);
            } else { // (the taps: a window of constant pitch — every row offset is an immediate of the read)
              const int j = lane & 3;
              const int slot0 = kTapRotate ? (256 * j + (lane & 48) + ((((lane >> 2) + j) & 3) << 2)) : (257 * j + (lane & ~3));
              s = win_tier_raw<CH, true>(win0 + slot0, 64, reinterpret_cast<const float *>(win0 + 1024) + slot0, psx - tx_, psy - ty_, after_reads);
            }
          } else {
            after_reads(); // nothing of this pass reads the buffer
            if constexpr (CH == 5) {
              const Px<5> s5 = sample_direct<2, Loop, 5, LRP_WIN_MINWAVES5 >= 4>(P, src, psx, psy);
              s = Rgba{s5.lo, s5.hi, s5.e};
            } else {
              s = sample_direct<2, Loop, 4, (LRP_WIN_MINWAVES >= 5), 4 * CH>(P, src, psx, psy);
            }
          }
          emit(g, k, s, std::integral_constant<bool, kRunsEverywhere>{}, !GeoRead || P.rgbaz_runs != 0);
          if constexpr (kPipeline) {
            behind_store = false;
            kind = next_kind;
            w = wn;
          }
        }
        if (!dma_early && has_next()) issue_next();
        if (last_frame) cur = nxt;
        continue;
      }
    }
    if (t_edge == 1 || t_edge == 2) edge_plane(cur);
#pragma unroll 2
    for (int h = 0; h < 2; ++h) {
      if (t_coef && (h == 0 || !t_whole)) precompute(cur, h);
      // (the RGBAZ big-window variant keeps the two passes of a half rolled — its code shrinks 82 -> 77 KB: rect -> equirect RGBAZ +
      // tonemap 262 -> 255 us batched; RGB / RGBA lose 0.5-1.5 % that way and stay unrolled)
      float hx0 = cur.sx[2 * h], hx1 = cur.sx[2 * h + 1], hy0 = cur.sy[2 * h], hy1 = cur.sy[2 * h + 1];
      if constexpr (kBigWin && CH == 5) asm volatile("" : "+v"(hx0), "+v"(hx1), "+v"(hy0), "+v"(hy1)); // (selected, not indexed: no scratch)
#pragma unroll(kBigWin && CH == 5 ? 1 : 2)
      for (int kk = 0; kk < 2; ++kk) {
        const int k = 2 * h + kk;
        if (Quad && !kSharedRays && k == 3 && last_frame && g + 1 < G) coords(g + 1, nxt); // only its box is kept
        if constexpr (GeoRead)
          if (k == 3 && g + 1 < G && last_frame) geo_plan(nxt, geo_class(g + 1), g + 1);
        const bool last_pass = k == 3;
        float psx = kk == 0 ? hx0 : hx1, psy = kk == 0 ? hy0 : hy1;
        if constexpr (Quad && !kSharedRays) quad_xy(image_of(g), k, psx, psy); // re-derived (2-4 instructions) instead of held in registers
        // (where the coordinates of a mirror image are a plain selection of stored values the compiler would otherwise
        // hoist everything derived from them — truncations, weights, window addresses of all four passes and both
        // images — out of the block loop and spill it: the selected values are opaque here)
        // (the same goes for the frame loop: everything derived from the coordinates of a pass is the same in every frame,
        // and kept for all four passes it does not fit the registers — what IS shared between frames is stage 1 and the
        // window plan, by construction)
        asm volatile("" : "+v"(psx), "+v"(psy));
        Rgba s;
        // behind the pixel's last read of the window: the next window's DMA (last pass), the second half of a split block (pass 1)
        auto after_reads = [&]() {
          if (last_pass) next_window();
          if (t_split && k == 1) issue(P.src, cur, 1);
        };
        if (t_edge != 0) {
          if (t_edge < 3) { // beyond the first / last source row
            const float tx_ = __builtin_truncf(psx);
            const int slot = (int)tx_ - 1 - cur.x_lo;
            s = win_tier_edge_row<CH>(win + cur.c_base + slot, reinterpret_cast<const float *>(win + cur.c_base + cur.bw) + slot, psx - tx_, after_reads);
          } else { // beyond the first / last source column
            const float ty_ = __builtin_truncf(psy);
            const int slot = (int)ty_ - 1 - cur.y_lo;
            s = win_tier_edge_col<CH>(win + slot, reinterpret_cast<const float *>(win + cur.bh) + slot, psy - ty_, t_edge == 4 ? 1.0f : 0.0f, after_reads);
          }
        } else if (t_coef) {
          const float tx_ = __builtin_truncf(psx), ty_ = __builtin_truncf(psy);
          // one 24-bit multiply per pixel (a 32-bit v_mul_lo_u32 issues at quarter rate); every other term
          // of the two addresses is wave-uniform and folded into tap_base / c_delta when the block is planned
          const int tap = __mul24((int)ty_, cur.spitch()) + (int)tx_ + cur.tap_base; // window slot of (int(sx) - 1, int(sy)): the second tap row
          s = win_tier_coef<CH>(win + tap, win + (tap + cur.c_delta(h)), cur.c_plane, cur.pitch,
                                reinterpret_cast<const float *>(win + cur.pitch * cur.bh) + (tap - cur.spitch()), psx - tx_, psy - ty_, after_reads);
        } else if (t_staged) {
          // (split blocks: h = 1 reads the window of passes 2-3, requested behind pass 1's taps — its arithmetic and store and
          // the other wavefronts cover part of the round trip — and waited for in front of pass 2)
          const int half = (t_split && h == 1) ? 1 : 0;
          if (t_split && k == 2) { // the DMA is older than every store behind it (SS: predicated stores — nothing is assumed)
            if constexpr (SS)
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else
              asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
          }
          const float tx_ = __builtin_truncf(psx), ty_ = __builtin_truncf(psy);
          const int slot0 = cur.org() + __mul24((int)ty_ - 1 - (kSplit ? cur.first_row_of(half) : cur.y_lo), cur.spitch()) + ((int)tx_ - 1 - cur.x_lo);
          s = win_tier_raw<CH>(win + slot0, cur.spitch(),
                               reinterpret_cast<const float *>(win + cur.pitch * (kSplit ? cur.rows_of(half) : cur.bh)) + slot0, psx - tx_, psy - ty_, after_reads);
        } else if (kRolledUnstaged) {
          __builtin_unreachable(); // (rendered by the rolled loop above)
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
  fill_share();
#if defined(LRP_WAVE_STAMPS)
  if (blockIdx.y == 0 && blockIdx.x < 65536u && (threadIdx.x & 63u) == 0) {
    const unsigned hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20); // HW_ID, XCC_ID
    g_wave_stamps[3 * blockIdx.x] = stamp_start;
    g_wave_stamps[3 * blockIdx.x + 1] = wall_clock64();
    g_wave_stamps[3 * blockIdx.x + 2] = (unsigned long long)hw_id | ((unsigned long long)(xcc_id & 15u) << 32) | ((unsigned long long)(unsigned)G << 40) | ((unsigned long long)(blockIdx.x % 8) << 60);
  }
#endif
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
    // ... and of the panorama sources, for geometries whose census says so (lrp_capi.cpp kBigWidePercent)
    if (big_windows && in_mode == kInEquirect) return reproject_bicubic_win_kernel<kEquirect, kInEquirect, 0, CH, false, true>;
    if (big_windows && in_mode == kInEquirectLoop) return reproject_bicubic_win_kernel<kEquirect, kInEquirectLoop, 0, CH, false, true>;
    // (round 4, without tap DMA: the pole face of the 8192^2 -> 2048^2 cubemap 108.6 -> 103.9 us, its side faces 59.8 -> 67.2, every
    // 4096^2 mapping out of a panorama 20-25 % slower; round 5 with tap DMA: pole face 97 -> 80, side faces 48 -> 60 — hence per geometry)
    return table[in_mode];
  }
};

// The supersampling instantiations (num_samples == 2): plain blocks, one per (output lens, source mode) cell.
template <int CH> struct WinSSKernelTable {
  static TileKernelFn get(int out_idx, int in_mode) {
    static const TileKernelFn table[3][4] = {
        {reproject_bicubic_win_kernel<kRect, kInRect, 0, CH, false, false, true>, reproject_bicubic_win_kernel<kRect, kInEquidistant, 0, CH, false, false, true>,
         reproject_bicubic_win_kernel<kRect, kInEquirect, 0, CH, false, false, true>, reproject_bicubic_win_kernel<kRect, kInEquirectLoop, 0, CH, false, false, true>},
        {reproject_bicubic_win_kernel<kEquidistant, kInRect, 0, CH, false, false, true>, reproject_bicubic_win_kernel<kEquidistant, kInEquidistant, 0, CH, false, false, true>,
         reproject_bicubic_win_kernel<kEquidistant, kInEquirect, 0, CH, false, false, true>, reproject_bicubic_win_kernel<kEquidistant, kInEquirectLoop, 0, CH, false, false, true>},
        {reproject_bicubic_win_kernel<kEquirect, kInRect, 0, CH, false, false, true>, reproject_bicubic_win_kernel<kEquirect, kInEquidistant, 0, CH, false, false, true>,
         reproject_bicubic_win_kernel<kEquirect, kInEquirect, 0, CH, false, false, true>, reproject_bicubic_win_kernel<kEquirect, kInEquirectLoop, 0, CH, false, false, true>}};
    return table[out_idx][in_mode];
  }
};

// ... and the ones that read an entry of sub-samples from the geometry cache: one per source mode.
template <int CH> struct WinSSGeoKernelTable {
  static TileKernelFn get(int in_mode) {
    static const TileKernelFn table[4] = {
        reproject_bicubic_win_kernel<kRect, kInRect, 0, CH, false, true, true>, reproject_bicubic_win_kernel<kRect, kInEquidistant, 0, CH, false, true, true>,
        reproject_bicubic_win_kernel<kRect, kInEquirect, 0, CH, false, true, true>, reproject_bicubic_win_kernel<kRect, kInEquirectLoop, 0, CH, false, true, true>};
    return table[in_mode];
  }
};

// num_samples must be 1 — or 2, 3, 4 for the SS launcher (a lane per sub-sample).
// QMode != 0: P.win_mode == QMode, set by the host only for cells where the mode exists.  GeoRead: P.geo_mode == 2, a single
// whole-image launch.
template <int QMode, int CH, bool GeoRead = false, bool SS = false>
inline hipError_t launch_win_bicubic_impl(KParams P, int out_idx, int in_mode, hipStream_t stream) {
  static_assert(!GeoRead || QMode == 0, "the geometry cache feeds plain blocks");
  static_assert(!SS || QMode == 0, "supersampling: plain blocks");
  if (GeoRead && (P.geo_mode != 2 || P.y_offset != 0 || P.y_end != P.out_h)) return hipErrorInvalidValue;
  if ((SS ? (P.num_samples < 2 || P.num_samples > 4) : P.num_samples != 1) || (SS && !GeoRead && P.geo_mode != 0 && P.geo_mode != 1)) return hipErrorInvalidValue;
  constexpr int kRowsPerBlock = SS ? kPassRows : kBlkH; // output rows of a block
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
    const int cols_per_block = SS ? 64 / (P.num_samples * P.num_samples) : kBlkW; // (SS: pixels of a pass — 16, 7, 4)
    P.tiles_x = (P.out_w + cols_per_block * kWinWaves - 1) / (cols_per_block * kWinWaves);
    // strips of LRP_WIN_STRIP blocks when that still leaves >= 8 workgroups per CU, else shorter
    const int row_blocks = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
#ifndef LRP_SS_STRIP
#define LRP_SS_STRIP 4
#endif
    int G = SS ? LRP_SS_STRIP : LRP_WIN_STRIP; // (SS: a block is four rows of pixels; strips of sixteen rows like everybody's)
    if (GeoRead && !SS && P.blocks_per_wave > 0) G = std::min(P.blocks_per_wave, kGeoStripRows); // the caller's override (lrp_debug_set "geo_strip")
    // (a batch whose wavefronts walk several frames pipelines the windows of one block across its frames: one block per
    // wavefront measured 2-3 % faster there — equirect -> fisheye rotated 143 -> 139 us, rect -> rect 130.5 -> 128 —, four 5 % slower)
    if (P.batch_n > 1 && !(out_idx == 2 && in_mode == kInRect)) G = 1;
    const bool strip_forced = GeoRead && !SS && P.blocks_per_wave > 0;
    // (a rectilinear source under a rectilinear / fisheye target: half of the blocks are corner blocks, the rest wait for
    // gathers or edge rows — single blocks balance the launch: rect -> fisheye single launches 164 -> 154 us)
    if (GeoRead && !SS && !strip_forced && in_mode == kInRect && P.big_windows == 0 && P.batch_n <= 1) G = 1;
    // >= 2 rounds of wavefronts; the launches that read the geometry cache >= 4 (their wavefronts differ more in what a block
    // costs them — nothing is computed, everything is waited for: pole face of the 8192^2 -> 2048^2 cubemap 107.5 -> 98.4 us)
    const long long min_waves = GeoRead ? 16384 : 8192;
    while (!strip_forced && G > 1 && (long long)P.tiles_x * kWinWaves * ((row_blocks + G - 1) / G) < min_waves) G >>= 1;
    if (GeoRead && P.geo_work != nullptr && in_mode != kInRect) return hipErrorInvalidValue; // (lists: the rectilinear source's instantiations)
    if (GeoRead && P.geo_work != nullptr) G = 1; // a listed launch: one block per wavefront, the blocks of the work list only
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
    if (GeoRead && P.big_windows != 0) F = 1; // (the big-window variant has no frame loop: a wavefront per block and frame)
    if (frames_override > 0) F = std::max(1, std::min(P.batch_n, frames_override)); // the caller's override (lrp_debug_set "batch_frames": A/B runs, tests)
    if (P.geo_mode == 1 || P.geo_mode == 3) F = 1; // the launch that writes a geometry-cache entry: the instantiations without the frame loop have the side output
    if (SS) F = 1;
    P.frames_per_wave = F;
    groups = (P.batch_n + F - 1) / F;
  }
  TileKernelFn fn;
  if constexpr (SS && GeoRead)
    fn = WinSSGeoKernelTable<CH>::get(in_mode);
  else if constexpr (SS)
    fn = WinSSKernelTable<CH>::get(out_idx, in_mode);
  else if constexpr (GeoRead)
    fn = P.frames_per_wave > 1 ? WinGeoFramesKernelTable<CH>::get(in_mode) : WinGeoKernelTable<CH>::get(in_mode, P.big_windows != 0);
  else
    fn = P.frames_per_wave > 1 ? WinKernelTable<QMode, CH, true>::get(out_idx, in_mode) : WinKernelTable<QMode, CH, false>::get(out_idx, in_mode);
  if (!fn) return hipErrorInvalidValue; // (the host never asks for a mode outside its cells)
  unsigned grid_x = (unsigned)(kXcds * xcd_rows(P.tiles_y, kWinXcdBand) * P.tiles_x);
  if (GeoRead && P.geo_work != nullptr) {
    if (kWinWaves != 1) return hipErrorInvalidValue; // (the work list names blocks, one per workgroup)
    if (P.geo_n_work == 0) return hipSuccess;       // every block is a corner block
    grid_x = P.geo_n_work;
  }
  hipLaunchKernelGGL(fn, dim3(grid_x, (unsigned)groups), dim3(kWinThreads), 0, stream, P);
  return hipGetLastError();
}

} // namespace lrp
