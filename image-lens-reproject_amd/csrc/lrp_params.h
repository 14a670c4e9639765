// lrp_params.h — kernel argument block shared by the C-ABI layer and the HIP
// kernels.  Plain POD passed by value in kernarg (SGPR-resident, wave-uniform).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lrp {

// Lens model ids used as template arguments (numbering of include/lrp.h /
// reference src/config.hpp:7-13).
enum : int { kRect = 0, kEquidistant = 1, kEquirect = 4 };
// Input-lens mode: equirectangular sources split into clamped and wrapping
// (reference LoopHorizontally, src/reproject.cpp:386-394).
enum : int { kInRect = 0, kInEquidistant = 1, kInEquirect = 2, kInEquirectLoop = 3 };
constexpr int kMaxBatch = 16; // frames of one geometry rendered by one launch (blockIdx.y = frame)
constexpr int kXcds = 8; // XCDs (private L2s) of an MI355X; blockIdx % 8 labels the blocks that share one
// XCD-aware tile numbering.  The dispatcher deals workgroup i to XCD i % 8.  The rows of tiles are cut
// into bands of kXcdBand rows and the bands dealt round-robin to the XCDs: XCD k walks bands k, k + 8,
// k + 16 ... in raster order, so the tiles one XCD has in flight (two rows of a 4K frame) neighbour each
// other and share source rows in its L2, while every XCD gets a share of every part of the frame.
// (One contiguous eighth of the frame per XCD — the round-1 numbering — leaves half the chip idle when
// the expensive tiles sit in one part of it: a rectilinear view inside an equirectangular panorama
// covers the four middle eighths only, a pole face the first or the last.)
#ifndef LRP_XCD_BAND
#define LRP_XCD_BAND 2
#endif
constexpr int kXcdBand = LRP_XCD_BAND;
// Rows of tiles one XCD walks (grid = 8 x this x tiles_x workgroups; surplus workgroups exit).
inline __host__ __device__ int xcd_rows(int tiles_y, int band = kXcdBand) {
  return (tiles_y + kXcds * band - 1) / (kXcds * band) * band;
}
// The window kernel deals single rows of blocks (measured against bands of two, 16-frame launches and single ones: window
// kernels 0-3 % faster — equirect -> rect 96.3 -> 94.7 us, RGBAZ fisheye -> rect 159.7 -> 155.8, cubemap pole face
// 113.2 -> 109.4 —, the bilinear tile kernel 1.5 % slower, which keeps its bands of two).
#ifndef LRP_WIN_XCD_BAND
#define LRP_WIN_XCD_BAND 1
#endif
constexpr int kWinXcdBand = LRP_WIN_XCD_BAND;

struct LensP {
  float p[4];          // union payload of LensInfo (see include/lrp.h)
  float sensor_width;
  float sensor_height;
};

struct KParams {
  const float *src;
  float *dst;
  int32_t in_w, in_h;
  int32_t out_w, out_h;
  int32_t channels;    // floats per texel (the stride)
  int32_t ch_count;    // pixel kernel, run-time channel path: channels rendered by this launch (<= 8; src / dst point at the first)
  int32_t num_samples;
  float normalize;     // 1.0f / (num_samples * num_samples), src/reproject.cpp:280
  LensP in_lens;
  LensP out_lens;
  float rot[9];        // row-major; valid when has_rot
  int32_t has_rot;
  int32_t has_post;    // fused post_process epilogue (src/reproject.cpp:421-437)
  float exposure;
  float reinhard;
  int32_t tiles_x, tiles_y; // output tiling of the launch
  int32_t y_offset;         // first output row of this launch (row-band launches)
  int32_t y_end;            // one past the last output row of this launch (out_h for a whole image)
  // ---- tile kernel (lrp_kernel_v2.h) only ----------------------------------
  // Separable output-lens terms, one entry per (column, sub-sample) and per
  // (row, sub-sample), built by lrp_tables.hip with the same operations the
  // per-pixel code would execute; null for the equidistant target (not separable).
  const float *col_tab; // [2][out_w * num_samples]
  const float *row_tab; // [out_h * num_samples]
  // Column-separable source x (lrp_tables.hip): rotated ray x, z and source texel x per
  // (column, sub-sample); null when the lens pair / rotation does not separate.
  const float *xsep_tab; // [3][out_w * num_samples]
  // Lens constants that depend on the lens only, evaluated once on the host
  // with the same IEEE binary32 operations (src/reproject.cpp:178,196,251-252,265-266).
  float in_focal, out_focal;        // equidistant: sensor_width / fov
  float in_lon_span, in_lat_span;   // equirectangular source
  int32_t blocks_per_wave;          // window kernel: 16 x 16 blocks per wavefront strip
  int32_t win_coef;                 // window kernel: shared tap-column coefficients allowed (0: raw taps only)
  int32_t alias_pairs;              // rectilinear view into a full-turn panorama, no pitch / roll: pixel (x + W/2, H-1-y) reads the texels of (x, y) (tile order, lrp_kernel_v2.h)
  // Batched launch (tile / window kernels): frame blockIdx.y reads batch_src[y], writes batch_dst[y]
  int32_t batch_n;
  const float *batch_src[kMaxBatch];
  float *batch_dst[kMaxBatch];
  int32_t quad;                     // 1: mirrored pixels / blocks (mapping symmetric about both image axes); 2: mirrored rays (tile kernels, equidistant target)
  int32_t frames_per_wave;          // window kernel, batched launch: consecutive frames one wavefront renders its strip for (blockIdx.y = group of frames)
  int32_t win_mode;                 // window kernel: 0 plain blocks, 1 blocks mirrored in both axes (== quad 1), 2 rows only (pan), 3 columns only (pitch), 4 shared rays (equidistant target, any rotation)
  int32_t win_edge;    // window kernel: blocks beyond one side of the source stage one source row / column (0: per-pixel gathers)
  int32_t win_split;   // window kernel: blocks whose window exceeds the buffer stage the windows of their two halves one after the other (0: per-pixel gathers)
  int32_t win_tapdma;  // window kernel: a pass whose window fits no buffer fetches its taps quad by quad through LDS-DMA (lrp_win_kernel.h tap_dma; 0: per-pixel gathers)
  // Geometry cache (lrp_geocache.h): what a single launch would re-derive from the geometry alone — the source
  // coordinates of every output pixel and, for the window kernel, the window extremes of every 16 x 16 block — kept in
  // HBM between calls.  geo_mode 0: not used; 1: this launch computes as usual and writes the entry as a side output;
  // 2: it loads instead of computing (the GeoRead instantiations; same values, hence the same bits).
  int32_t geo_mode;
  float *geo_xy;       // (sx, sy) of every output pixel, element geo_map_index(x, y): the top-left-origin source texel coordinates of src/reproject.cpp:323-324
  int32_t *geo_box;    // window kernel: [block rows][blocks_x][8] (geo_layout below)
  int32_t big_windows; // GeoRead window kernel: the big-window variant (20 KiB of LDS per wavefront, two wavefronts per SIMD) where it is instantiated
  int32_t rgbaz_runs;  // GeoRead window kernel: a rectilinear view rendered into a panorama — the big-window variant; RGBAZ: every block leaves as 16-byte chunks (what the compute instantiations of that mapping do at compile time)
  // Block lists of a geometry-cache entry (GeoLists below, lrp_geo_lists.hip).  geo_work != null: a GeoRead window launch
  // walks the WORK list — workgroup i renders block geo_work[i] = (column, row) of 16 x 16 blocks, or exits on a -1 —
  // instead of enumerating the frame; the corner blocks, which are not on it, are written by the fill kernel from geo_runs.
  const int32_t *geo_work;   // [geo_n_work][2]
  const int32_t *geo_work_rec; // [geo_n_work][8]: the box records of the work list's blocks (null: loaded from the box array per block)
  const uint32_t *geo_runs;  // [geo_n_runs][4]: block row, first block column, blocks (<= kGeoRunBlocks), corner class 1-4
  uint32_t geo_n_work, geo_n_runs;
  uint32_t geo_fill_per_wave; // listed window launch: row segments of the corner runs a filling wavefront writes when it is done (0: the fill kernel writes them)
  uint32_t geo_fill_stride;   // ... every geo_fill_stride-th wavefront fills (odd: the filling wavefronts then fall on all XCDs): a share of a whole run or more per filling wavefront — its set-up (run record, corner texel, tonemap: two scalar round trips) is paid once per share
};

// A geometry-cache entry (num_samples == 1, whole images): the coordinate map and, for the window kernel, per 16 x 16
// block (block row * blocks_x + block column) 8 words — lo_x, hi_x, lo_y / hi_y of passes 0-1, lo_y / hi_y of passes 2-3
// (float bits of the wave-wide extremes of the block's coordinates), flags (bit 0 / 1: the taps of every pixel are
// consecutive in x / y, bit 2: a window was planned from these), 0.
inline __host__ __device__ uint32_t geo_block_cols(int out_w) { return (uint32_t)(out_w + 15) / 16; }
constexpr int kGeoStripRows = 16; // block rows are allocated in multiples of the longest strip a GeoRead launch may walk
struct GeoLayout {
  size_t xy_bytes, box_bytes, list_bytes;
  size_t bytes() const { return xy_bytes + box_bytes + list_bytes; }
};
// Block lists, behind the class bytes (built once per entry by geo_build_lists, lrp_geo_lists.hip, from the class bytes):
//   header  kGeoListHeaderWords words: [0] work entries, [1] runs, [2] corner blocks, [3] blocks of the image, [4], [5] unused
//           [6] blocks in view whole that no 10 KiB window stages, [7] blocks in view whole (the census, lrp_geo_lists.hip)
//   work    pairs (block column, block row) of every block that is NOT a corner block, in the order the window kernel's
//           launch would reach them: entry i goes to workgroup i, i.e. to XCD i % 8, and the entries of one XCD are its
//           rows of blocks (row % 8 == XCD) in raster order with the corner blocks taken out (alias pairs stay neighbours);
//           sub-lists shorter than the longest end in (-1, -1).  Dense and cost-uniform: no wave slot is spent on stores only.
//   runs    maximal runs of horizontally adjacent corner blocks of one class inside an aligned group of kGeoRunBlocks
//           block columns: (block row, first block column, blocks, class).  Every pixel of a run is the same value
//           (the clamped corner texel, src/reproject.cpp:114-131), so a run is 16 contiguous row segments of that value.
//   recs    the box record (8 words) of every entry of the work list, at the entry's position: a listed wavefront reads its block
//           and the block's record with two scalar loads issued together — one round trip in front of the window request instead
//           of two (the record's address in the box array depends on the entry).
constexpr int kGeoListHeaderWords = 16;
constexpr int kGeoRunBlocks = 16;
inline __host__ __device__ uint32_t geo_image_block_rows(int out_h) { return (uint32_t)(out_h + 15) / 16; }
inline __host__ __device__ size_t geo_work_capacity(int out_w, int out_h) { // entries (pairs)
  return (size_t)kXcds * ((geo_image_block_rows(out_h) + kXcds - 1) / kXcds) * geo_block_cols(out_w);
}
inline __host__ __device__ size_t geo_run_capacity(int out_w, int out_h) { return (size_t)geo_image_block_rows(out_h) * geo_block_cols(out_w); }
inline __host__ __device__ size_t geo_work_recs_offset(int out_w, int out_h) { // bytes from the list header to the records of the work list
  return (size_t)kGeoListHeaderWords * 4 + geo_work_capacity(out_w, out_h) * 8 + geo_run_capacity(out_w, out_h) * 16;
}
// Element (float2) of output pixel (x, y) in the coordinate map: row-major.  (A map stored in 16 x 16 tiles — 2 KiB contiguous
// bytes per block of the window kernel instead of 16 row segments of 128 bytes — measured the same for the window kernels
// and 2-3 % slower for the tile kernels: profiles/r04_experiments_ab.txt.)
inline __host__ __device__ uint32_t geo_map_index(int x, int y, int out_w) { return (uint32_t)y * (uint32_t)out_w + (uint32_t)x; }
// Block rows an entry allocates, and — behind the box records — one CLASS byte per block, stored column by column
// (block column * geo_block_rows + block row: the blocks of a wavefront's strip are consecutive bytes of one aligned word):
// 0, or 1-4 = every pixel of the block lies beyond that corner of the source (WinBlockT::corner()).  A GeoRead wavefront reads
// the classes of its strip with one scalar load when it starts; a corner block then needs neither its box record nor its
// coordinates — no vector load at all in front of its four stores.
inline __host__ __device__ uint32_t geo_block_rows(int out_h) {
  const uint32_t by = (uint32_t)(out_h + 15) / 16;
  return (by + kGeoStripRows - 1) / kGeoStripRows * kGeoStripRows;
}
inline __host__ __device__ size_t geo_class_offset(int out_w, int out_h) { // bytes from the first box record to the first class byte
  return (size_t)geo_block_cols(out_w) * geo_block_rows(out_h) * 32;
}
inline __host__ __device__ size_t geo_lists_offset(int out_w, int out_h) { // bytes from the first box record to the list header
  const size_t blocks = (size_t)geo_block_cols(out_w) * geo_block_rows(out_h);
  return (blocks * 32 + blocks + 255) & ~(size_t)255;
}
// num_samples > 1 (the supersampling instantiations of the window kernel): the map holds one pair per SUB-SAMPLE, the ns^2 of a
// pixel next to each other in the reference's order — element geo_ss_map_index — and nothing else (a block's window is planned
// from the loaded coordinates).
inline __host__ __device__ uint32_t geo_ss_map_index(int x, int y, int out_w, int ns2, int sub) {
  return ((uint32_t)y * (uint32_t)out_w + (uint32_t)x) * (uint32_t)ns2 + (uint32_t)sub;
}
inline GeoLayout geo_layout(int out_w, int out_h, bool with_boxes, int num_samples = 1) {
  GeoLayout L{};
  L.xy_bytes = ((size_t)out_w * (size_t)out_h * (size_t)(num_samples * num_samples) * 8 + 255) & ~(size_t)255; // (the records and lists behind the map start on a 256-byte boundary)
  if (with_boxes && num_samples == 1) {
    L.box_bytes = geo_lists_offset(out_w, out_h);
    L.list_bytes = geo_work_recs_offset(out_w, out_h) + geo_work_capacity(out_w, out_h) * 32;
  }
  return L;
}

} // namespace lrp
