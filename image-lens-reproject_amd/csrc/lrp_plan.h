// lrp_plan.h — the launch planner: every decision between "the caller asked for this reprojection" and "this kernel family,
// with these sharing / staging paths, reading or writing the geometry cache this way" as PURE functions of the request, the
// switches and the facts the host learns on the way (what the output-lens tables look like, what the geometry cache holds).
// No HIP, no global state, no pointers: lrp_capi.cpp (enqueue_reproject) asks the planner and then launches; the CPU tests
// (tests/native/plan_driver.cpp, tests/test_plan.py) ask it the same questions without a GPU, so a changed threshold or rule
// shows up in a table-driven test.  The stages follow the order in which the facts become known:
//
//   plan_family   request + switches                      -> tile / window family at all? does it want the output-lens tables?
//   plan_rotation + facts about the tables                -> is the identity matrix dropped? is the column-separable x table wanted?
//   plan_sharing  + whether that table exists             -> mirror modes, window-kernel tiers, alias pairs, and whether the
//                                                            launch goes to the geometry cache (and as which kind of user)
//   plan_geo      + what geo_acquire answered             -> GeoRead variant (big windows by the census), rendering by block
//                                                            class (lists), the fused corner fill's shares
//
// Thresholds were measured on MI355X boxes (profiles/r0N_experiments_ab.txt, tools/policy_check.py re-times every automatic
// choice against its alternatives on the box it runs on).
#pragma once
#include <cstdint>

namespace lrp {

// Numbering of lrp_params.h (static_asserts in lrp_capi.cpp keep the two in step).
enum PlanLens : int { kPlanRect = 0, kPlanEquidistant = 1, kPlanEquirect = 4 };
enum PlanInMode : int { kPlanInRect = 0, kPlanInEquidistant = 1, kPlanInEquirect = 2, kPlanInEquirectLoop = 3 };
enum PlanInterp : int { kPlanNearest = 0, kPlanBilinear = 1, kPlanBicubic = 2 };

// Listed launches render one block per wavefront, the enumerating launch strips of two (its wavefronts fetch the next block's
// record under the current block): with the corner runs written by every n-th wavefront the listed launch is level at a third of
// a frame of corner blocks (BASELINE configs[3], 37 %: RGBA +-1 %, RGBAZ + tonemap 2-4 % ahead) and ahead beyond that (rect ->
// fisheye 2-3 %, narrower views 10-40 %); below, the plain enumeration stays.
constexpr unsigned kListedCornerPercent = 30;
// in-view blocks no 10 KiB window stages, per cent: from there on the big-window variant renders a panorama source
constexpr unsigned kBigWidePercent = 30;
// wavefronts a listed window launch must have to carry the corner runs itself
constexpr unsigned kMinWavesForFusedFill = 2048;
// largest num_samples the window kernel's supersampling instantiations take (sub-sample loop; 5 and more: the tile kernel)
constexpr int kMaxWindowSamples = 4;

// The switches of lrp_debug_set that decide anything here, at their current values (lrp_capi.cpp reads them once per call).
struct PlanSwitches {
  int kernel = 2;        // 0 pixel kernel, 1 tile kernels only, 2 tile + window kernels, 3 the same without any work sharing
  int xsep = 1, quad = 1, mirror_modes = 1, win_edge = 1, win_split = 1, win_tapdma = 1, win_ss = 1;
  int batch_frames = 0;  // frames per wavefront of a batched launch (0: the launcher decides)
  int geo_cache = 1, geo_strip = 0, geo_big = 1, geo_lists = 1, geo_fill_fused = 1, geo_list_recs = 1;
};

struct PlanRequest {
  int out_type = kPlanRect;    // PlanLens of the output lens
  int in_type = kPlanRect;     // PlanLens of the input lens
  int in_mode = kPlanInRect;   // PlanInMode (equirectangular sources: clamped / wrapping, src/reproject.cpp:386-394)
  int out_w = 0, out_h = 0, in_w = 0, in_h = 0, channels = 0;
  int num_samples = 1, interpolation = kPlanBicubic;
  bool has_rot = false;
  float rot[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  float out_lon_span = 0.0f;   // equirectangular target: longitude_max - longitude_min (alias pairs need the full turn)
  int n_batch = 0;             // > 0: a batch of that many frames of one geometry
  bool band = false;           // a row band of the output, not the whole image
  bool byte_offsets_fit = true; // both images below 4 GiB (the tile / window kernels address through 32-bit byte offsets)
};

struct PlanFamily {
  bool tile = false;         // tile / window kernels (else: one pixel per lane, any channel count, any size)
  bool wants_tables = false; // separable output-lens terms (rectilinear / equirectangular target)
};

// What get_output_tables reported (lrp_tables.h): `built` false = no memory for them (the pixel kernel renders instead).
struct TableFacts {
  bool built = false;
  bool plain = false; // every ray component finite, no -0.0f
  int symmetry = 0;   // bit 0 / 1: the column / row terms are mirror images about the image centre, bit for bit
};

struct PlanRotation {
  bool has_rot = false;    // after an exact identity matrix has been dropped
  bool wants_xsep = false; // the column-separable source-x table (rotation rows 0 / 2 do not see vy)
};

struct PlanSharing {
  int quad = 0;            // KParams::quad
  bool window = false;     // the LDS-window kernel renders (bicubic, RGB / RGBA / RGBAZ, num_samples 1 or the SS instantiations)
  bool window1 = false;    // ... with one sample per pixel
  int win_mode = 0, win_coef = 0, win_edge = 0, win_split = 0, win_tapdma = 0;
  int alias_pairs = 0;
  int frames_per_wave = 0;
  bool wants_geo = false;      // this launch goes to the geometry cache (geo_acquire)
  bool geo_want_boxes = false; // ... as a window launch (needs the per-block records too)
};

// What geo_acquire answered (lrp_geocache.h GeoUse).
struct GeoFacts {
  int mode = 0; // 0 no cache for this launch, 1 write map + records, 3 write records, 2 read
  bool lists = false;
  uint32_t n_work = 0, n_runs = 0, n_corner_blocks = 0, n_blocks = 0, n_wide = 0, n_inview = 0;
};

struct PlanGeo {
  int geo_mode = 0;
  int win_mode = 0, quad = 0; // (a launch that uses the cache renders plain blocks)
  int blocks_per_wave = 0, rgbaz_runs = 0, big_windows = 0;
  bool listed = false;        // rendering by block class: the window kernel walks the work list
  bool list_recs = false;     // ... and reads its block's record from beside the list entry
  uint32_t fill_stride = 0, fill_per_wave = 0; // the corner runs as a share per wavefront (0: the fill kernel writes them)
};

PlanFamily plan_family(const PlanRequest &r, const PlanSwitches &s);
PlanRotation plan_rotation(const PlanRequest &r, const PlanSwitches &s, const PlanFamily &f, const TableFacts &t);
PlanSharing plan_sharing(const PlanRequest &r, const PlanSwitches &s, const PlanFamily &f, const TableFacts &t, const PlanRotation &rot,
                         bool xsep_available);
PlanGeo plan_geo(const PlanRequest &r, const PlanSwitches &s, const PlanSharing &sh, const GeoFacts &g);

} // namespace lrp
