// lrp_capi.cpp — the C ABI declared in include/lrp.h: argument validation in the
// reference's dispatch order, kernel-argument construction, device buffers,
// streams and the batch context.  Host code only; the kernels live in the
// lrp_kernels_*.hip / lrp_aux_kernels.hip translation units.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/lrp.h"
#include "lrp_geocache.h"
#include "lrp_params.h"
#include "lrp_plan.h"
#include "lrp_tables.h"

namespace lrp {
hipError_t launch_nearest(const KParams &P, int out_idx, int in_mode, hipStream_t stream);
hipError_t launch_bilinear(const KParams &P, int out_idx, int in_mode, hipStream_t stream);
hipError_t launch_bicubic(const KParams &P, int out_idx, int in_mode, hipStream_t stream);
hipError_t launch_tile_nearest(const KParams &P, int out_idx, int in_mode, hipStream_t stream);
hipError_t launch_tile_bilinear(const KParams &P, int out_idx, int in_mode, hipStream_t stream);
hipError_t launch_tile_bicubic(const KParams &P, int out_idx, int in_mode, hipStream_t stream);
hipError_t launch_win_bicubic(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // (P.geo_mode == 2: the GeoRead kernels)
hipError_t launch_ss_gather(const KParams &P, int interpolation, int in_mode, hipStream_t stream); // lrp_tile_ssg.hip: nearest / bilinear, num_samples 2-4, from an entry of sub-samples
hipError_t launch_geo_build_lists(int32_t *box, int out_w, int out_h, int alias_pairs, hipStream_t stream); // lrp_geo_lists.hip
hipError_t launch_geo_census(int32_t *box, int out_w, int out_h, int in_w, int in_h, bool clear_header, hipStream_t stream); // lrp_geo_lists.hip
hipError_t launch_corner_fill(const KParams &P, hipStream_t stream);
hipError_t launch_post_process(float *data, uint32_t n_pixels, int channels, float exposure, float reinhard,
                               hipStream_t stream);
hipError_t launch_synth_fill(float *data, uint32_t n_elems, int channels, uint32_t seed, int depth_channel,
                             hipStream_t stream);
hipError_t launch_math_eval(int func, const float *a, const float *b, float *out, size_t n, hipStream_t stream);
hipError_t launch_checksum(const float *data, size_t n, unsigned long long *out, hipStream_t stream);
size_t pixel_bytes(int format, int channels);
hipError_t launch_decode_pixels(const void *src, int format, int src_channels, float *dst, int dst_channels,
                                size_t n_pixels, int device, hipStream_t stream);
hipError_t launch_encode_pixels(const float *src, int src_channels, void *dst, int format, int dst_channels,
                                unsigned fill, size_t n_pixels, int device, hipStream_t stream);
void pixel_tables_host(float decode[256], float threshold[256]);
} // namespace lrp

namespace {

thread_local std::string g_last_error;

int fail(int status, const std::string &detail) {
  g_last_error = detail;
  return status;
}

int hip_fail(hipError_t e, const char *what) {
  int st = (e == hipErrorOutOfMemory) ? LRP_ERR_OOM : LRP_ERR_HIP;
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) st = LRP_ERR_NO_DEVICE;
  return fail(st, std::string(what) + ": " + hipGetErrorString(e));
}

#define LRP_HIP_TRY(expr)                                                                                    \
  do {                                                                                                       \
    hipError_t _e = (expr);                                                                                  \
    if (_e != hipSuccess) return hip_fail(_e, #expr);                                                        \
  } while (0)

int device_count_cached() {
  static int count = [] {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    return n < 0 ? 0 : n;
  }();
  return count;
}

int select_device(int device) {
  const int n = device_count_cached();
  if (n == 0) return fail(LRP_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU path");
  if (device < 0 || device >= n) return fail(LRP_ERR_NO_DEVICE, "device index out of range");
  LRP_HIP_TRY(hipSetDevice(device));
  return LRP_OK;
}

bool lens_in_hot_path(int type) {
  return type == LRP_RECTILINEAR || type == LRP_FISHEYE_EQUIDISTANT || type == LRP_EQUIRECTANGULAR;
}

int out_lens_index(int type) { return type == LRP_RECTILINEAR ? 0 : (type == LRP_FISHEYE_EQUIDISTANT ? 1 : 2); }

// LoopHorizontally decision, reference src/reproject.cpp:386-394: float span,
// compared in double against 2*M_PI with a float threshold.
bool source_wraps(const lrp_lens &L) {
  const float long_range = L.u.equirectangular.longitude_max - L.u.equirectangular.longitude_min;
  return std::fabs((double)long_range - (2 * M_PI)) < 1e-5f;
}

int in_lens_mode(const lrp_lens &L) {
  if (L.type == LRP_RECTILINEAR) return lrp::kInRect;
  if (L.type == LRP_FISHEYE_EQUIDISTANT) return lrp::kInEquidistant;
  return source_wraps(L) ? lrp::kInEquirectLoop : lrp::kInEquirect;
}

lrp::LensP pack_lens(const lrp_lens &L) {
  lrp::LensP p;
  std::memcpy(p.p, L.u.raw, sizeof(p.p));
  p.sensor_width = L.sensor_width;
  p.sensor_height = L.sensor_height;
  return p;
}

size_t image_bytes(const lrp_image &im) { return (size_t)im.width * (size_t)im.height * (size_t)im.channels * 4u; }
// The reference addresses texels with `int` (src/reproject.cpp:49-51,134-142: ly * pitch + lx * channels + c): an image of
// up to 2^31 floats (8 GiB) is what it can render; beyond that its index overflows.  Same limit here.
constexpr unsigned long long kMaxImageFloats = 1ull << 31;
bool image_addressable(const lrp_image &im) {
  return (unsigned long long)im.width * (unsigned long long)im.height * (unsigned long long)im.channels <= kMaxImageFloats;
}
// The tile / window kernels address texels through 32-bit byte offsets (buffer descriptors, LDS-DMA): images below 4 GiB.
// Larger ones — [4 GiB, 8 GiB] — take the one-pixel-per-lane kernel, whose element offsets are 32-bit and pointers 64-bit.
bool image_fits_byte_offsets(const lrp_image &im) { return (unsigned long long)image_bytes(im) < (1ull << 32); }

// Checks in the order the reference dispatches: output lens
// (src/reproject.cpp:408-418), input lens (:378-398), interpolation (:352-367);
// then the preconditions the reference leaves unchecked.
int validate(const lrp_image *in, const lrp_image *out, int interpolation, bool need_data) {
  if (!in || !out) return fail(LRP_ERR_NULL, "null image");
  if (!lens_in_hot_path(out->lens.type)) return fail(LRP_ERR_OUTPUT_LENS, "Output lens type not supported.");
  if (!lens_in_hot_path(in->lens.type)) return fail(LRP_ERR_INPUT_LENS, "Input lens type not supported.");
  if (interpolation != LRP_NEAREST && interpolation != LRP_BILINEAR && interpolation != LRP_BICUBIC)
    return fail(LRP_ERR_INTERPOLATION, "Interpolation method not supported.");
  if (in->channels < 1 || in->channels != out->channels)
    return fail(LRP_ERR_CHANNELS, "in->channels must equal out->channels and be >= 1");
  if (in->width < 1 || in->height < 1 || out->width < 1 || out->height < 1)
    return fail(LRP_ERR_BAD_DIMS, "image dimensions must be positive");
  if (!image_addressable(*in) || !image_addressable(*out))
    return fail(LRP_ERR_BAD_DIMS, "an image of more than 2^31 floats (8 GiB) cannot be addressed (the reference indexes texels with int, src/reproject.cpp:49-51)");
  if (need_data && (!in->data || !out->data)) return fail(LRP_ERR_NULL, "null image data");
  return LRP_OK;
}

lrp::KParams make_params(const lrp_image *in, const lrp_image *out, int num_samples, const float *rotation,
                         const lrp_post *post) {
  lrp::KParams P;
  std::memset(&P, 0, sizeof(P));
  P.src = in->data;
  P.dst = out->data;
  P.in_w = in->width;
  P.in_h = in->height;
  P.out_w = out->width;
  P.out_h = out->height;
  P.channels = out->channels;
  P.ch_count = out->channels;
  P.num_samples = num_samples;
  P.normalize = 1.0f / (float)(num_samples * num_samples); // src/reproject.cpp:280
  P.in_lens = pack_lens(in->lens);
  P.out_lens = pack_lens(out->lens);
  P.has_rot = rotation != nullptr;
  if (rotation) std::memcpy(P.rot, rotation, sizeof(P.rot));
  P.has_post = post != nullptr;
  if (post) {
    P.exposure = post->exposure;
    P.reinhard = post->reinhard;
  }
  P.y_offset = 0;
  P.y_end = out->height;
  // lens-only constants of the tile kernel, same binary32 operations as the
  // reference performs per pixel (src/reproject.cpp:178,196,265-266)
  P.in_focal = in->lens.sensor_width / in->lens.u.fisheye_equidistant.fov;
  P.out_focal = out->lens.sensor_width / out->lens.u.fisheye_equidistant.fov;
  P.in_lon_span = in->lens.u.equirectangular.longitude_max - in->lens.u.equirectangular.longitude_min;
  P.in_lat_span = in->lens.u.equirectangular.latitude_max - in->lens.u.equirectangular.latitude_min;
  return P;
}

// Kernel selection and the A/B switches of the sharing / staging paths.  All of them live in one table of atomics that
// lrp_debug_set() reads and writes (tests, tools/policy_check.py); the library reads no environment variable.
//   kernel: 0 = pixel kernel, 1 = tile kernel everywhere, 2 (default) = tile kernel with the LDS-window kernel for
//   bicubic, 3 = the same without its shared-coefficient tier and without any work sharing.  All HIP; there is no CPU path.
enum DebugKnob : int { kKnobKernel = 0, kKnobXsep, kKnobQuad, kKnobMirrorModes, kKnobWinEdge, kKnobWinSplit, kKnobBatchFrames, kKnobMultiFork, kKnobGeoCache, kKnobGeoStrip, kKnobGeoBig, kKnobGeoLists, kKnobGeoFillStream, kKnobGeoFillFused, kKnobContextStreams, kKnobWinTapDma, kKnobGeoCensus, kKnobGeoListRecs, kKnobWinSS, kKnobListedLaunches, kKnobBigLaunches, kKnobCount };
struct KnobSpec {
  const char *name;
  int lo, hi, initial;
};
constexpr int kMaxSideStreams = 5;
const KnobSpec kKnobs[kKnobCount] = {
    {"kernel", 0, 3, 2},
    {"xsep", 0, 1, 1},                  // column-separable source x tables
    {"quad", 0, 1, 1},                  // mirrored pixels / blocks (every mirror mode)
    {"mirror_modes", 0, 1, 1},  // window kernel: pan / pitch / shared-ray mirror modes
    {"win_edge", 0, 1, 1},          // window kernel: blocks beyond one side of the source stage one row / column
    {"win_split", 0, 1, 1},        // window kernel: split blocks and pass windows
    {"batch_frames", 0, lrp::kMaxBatch, 0}, // frames per wavefront of a batched launch (0: automatic)
    {"multi_fork", 0, kMaxSideStreams, 1},    // side streams of lrp_reproject_multi_device
    {"geo_cache", 0, 1, 1},        // geometry cache used by single launches (0: every launch computes)
    {"geo_strip", 0, lrp::kGeoStripRows, 0}, // blocks per wavefront of a launch that reads the geometry cache (0: automatic)
    {"geo_big", 0, 2, 1},            // big-window variant of the kernels that read the geometry cache (a rectilinear view rendered into a panorama); 0: the four-wavefront instantiation
    {"geo_lists", 0, 2, 1},        // rendering by block class from the lists of a geometry-cache entry (corner runs by the fill kernel / a share per wavefront, the window kernel over the work list): 0 never, 1 where at least 30 % of the blocks are corner blocks, 2 whenever the lists are known
    {"geo_fill_stream", 0, 1, 0}, // the fill kernel of a listed launch: 0 in front of the window kernel on the caller's stream, 1 beside it on a side stream of the device
    {"geo_fill_fused", 0, 1, 1}, // the corner runs of a listed launch as a share per wavefront of the window kernel (0: always the fill kernel)
    {"context_streams", 0, 1, 1}, // lrp_context: consecutive images alternate between two compute streams (0: one)
    {"win_tapdma", 0, 1, 1},      // window kernel: passes whose window fits no buffer fetch their taps a quad of lanes per pixel row through LDS-DMA (0: a gather per lane and tap)
    {"geo_census", 0, 1, 1},      // the census of a new geometry-cache entry's windows (lrp_geo_lists.hip; what the automatic choice of the big-window variant reads); 0: not taken
    {"geo_list_recs", 0, 1, 1}, // listed launches: a wavefront reads its block's box record from beside its work-list entry, with the entry (0: from the box array, a second round trip)
    {"win_ss", 0, 1, 1},              // bicubic with num_samples == 2 through the window kernel's supersampling instantiations (0: the tile kernel, as for any other num_samples > 1)
    {"listed_launches", 0, 0, 0}, // a counter, not a switch: launches rendered by block class so far (set 0 to reset; tests, bench)
    {"big_launches", 0, 0, 0},       // a counter: window launches through the big-window variant so far
};
std::atomic<int> g_knobs[kKnobCount];
const bool g_knobs_initialised = [] {
  for (int k = 0; k < kKnobCount; ++k) g_knobs[k].store(kKnobs[k].initial, std::memory_order_relaxed);
  return true;
}();
int knob(int k) { return g_knobs[k].load(std::memory_order_relaxed); }
int kernel_choice() { return knob(kKnobKernel); }

// The fill kernel of a listed launch (lrp_geo_lists.hip).  In front of the window kernel on the caller's stream, or — knob
// "geo_fill_stream" — beside it on a side stream of the device: the two write disjoint pixels, the window kernel holds
// two to four wavefronts per SIMD and all of a CU's LDS, the fill kernel needs neither.  Fork / join by events, so the
// caller's stream order is kept; not while that stream is being captured.
struct FillFork {
  std::mutex busy; // one fork / join being enqueued at a time per device (the events are re-recorded by every call)
  hipStream_t side = nullptr;
  hipEvent_t forked = nullptr, joined = nullptr;
};
std::mutex g_fill_fork_mutex;
std::map<int, std::unique_ptr<FillFork>> g_fill_forks;
FillFork *fill_fork(int device) {
  std::lock_guard<std::mutex> lock(g_fill_fork_mutex);
  std::unique_ptr<FillFork> &slot = g_fill_forks[device];
  if (!slot) {
    std::unique_ptr<FillFork> f(new FillFork);
    if (hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&f->forked, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&f->joined, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    slot = std::move(f);
  }
  return slot.get();
}
hipError_t fill_corner_runs(const lrp::KParams &P, int device, hipStream_t stream) {
  if (P.geo_n_runs == 0) return hipSuccess;
  if (knob(kKnobGeoFillStream) != 0) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
    FillFork *const f = cap == hipStreamCaptureStatusNone ? fill_fork(device) : nullptr;
    if (f != nullptr) {
      std::lock_guard<std::mutex> lock(f->busy);
      if (hipEventRecord(f->forked, stream) == hipSuccess && hipStreamWaitEvent(f->side, f->forked, 0) == hipSuccess) {
        const hipError_t e = lrp::launch_corner_fill(P, f->side);
        // join, whatever happened: the caller's stream continues behind the side stream
        if (hipEventRecord(f->joined, f->side) != hipSuccess || hipStreamWaitEvent(stream, f->joined, 0) != hipSuccess) {
          (void)hipGetLastError();
          (void)hipStreamSynchronize(f->side);
        }
        return e;
      }
      (void)hipGetLastError();
    }
  }
  return lrp::launch_corner_fill(P, stream);
}

// n_batch > 0: `in` / `out` are arrays of n_batch images of one geometry (checked by the caller);
// the tile / window kernels render them in launches of up to kMaxBatch frames, the per-pixel
// kernels one launch per frame.
// row_count > 0: only output rows [row_first, row_first + row_count) are rendered (the rows of the reference
// loop are independent, src/reproject.cpp:284); the kernels that share work between mirrored rows need the
// whole image and are not used for a band.
// The switches the planner reads, at their current values.
lrp::PlanSwitches plan_switches() {
  lrp::PlanSwitches s;
  s.kernel = kernel_choice();
  s.xsep = knob(kKnobXsep), s.quad = knob(kKnobQuad), s.mirror_modes = knob(kKnobMirrorModes);
  s.win_edge = knob(kKnobWinEdge), s.win_split = knob(kKnobWinSplit), s.win_tapdma = knob(kKnobWinTapDma), s.win_ss = knob(kKnobWinSS);
  s.batch_frames = knob(kKnobBatchFrames);
  s.geo_cache = knob(kKnobGeoCache), s.geo_strip = knob(kKnobGeoStrip), s.geo_big = knob(kKnobGeoBig), s.geo_lists = knob(kKnobGeoLists);
  s.geo_fill_fused = knob(kKnobGeoFillFused), s.geo_list_recs = knob(kKnobGeoListRecs);
  return s;
}

lrp::PlanRequest plan_request(const lrp_image *in, const lrp_image *out, int num_samples, int interpolation, const float *rotation,
                              int n_batch, bool band) {
  lrp::PlanRequest r;
  r.out_type = out->lens.type == LRP_RECTILINEAR ? lrp::kPlanRect : (out->lens.type == LRP_FISHEYE_EQUIDISTANT ? lrp::kPlanEquidistant : lrp::kPlanEquirect);
  r.in_type = in->lens.type == LRP_RECTILINEAR ? lrp::kPlanRect : (in->lens.type == LRP_FISHEYE_EQUIDISTANT ? lrp::kPlanEquidistant : lrp::kPlanEquirect);
  r.in_mode = in_lens_mode(in->lens);
  r.out_w = out->width, r.out_h = out->height, r.in_w = in->width, r.in_h = in->height, r.channels = out->channels;
  r.num_samples = num_samples, r.interpolation = interpolation;
  r.has_rot = rotation != nullptr;
  if (rotation) std::memcpy(r.rot, rotation, sizeof(r.rot));
  r.out_lon_span = out->lens.u.equirectangular.longitude_max - out->lens.u.equirectangular.longitude_min;
  r.n_batch = n_batch;
  r.band = band;
  r.byte_offsets_fit = image_fits_byte_offsets(*in) && image_fits_byte_offsets(*out);
  return r;
}
static_assert((int)lrp::kPlanRect == (int)lrp::kRect && (int)lrp::kPlanEquidistant == (int)lrp::kEquidistant && (int)lrp::kPlanEquirect == (int)lrp::kEquirect,
              "lrp_plan.h numbers lenses like lrp_params.h");
static_assert((int)lrp::kPlanInRect == (int)lrp::kInRect && (int)lrp::kPlanInEquidistant == (int)lrp::kInEquidistant &&
                  (int)lrp::kPlanInEquirect == (int)lrp::kInEquirect && (int)lrp::kPlanInEquirectLoop == (int)lrp::kInEquirectLoop,
              "lrp_plan.h numbers input modes like lrp_params.h");
static_assert((int)lrp::kPlanNearest == LRP_NEAREST && (int)lrp::kPlanBilinear == LRP_BILINEAR && (int)lrp::kPlanBicubic == LRP_BICUBIC, "interpolation numbering");

// The launcher: asks the planner (lrp_plan.h — every decision is there, as pure functions the CPU tests call too), fetches what
// the plan wants (output-lens tables, the column-separable x table, the geometry-cache entry) and enqueues the launches.
int enqueue_reproject(const lrp_image *in, lrp_image *out, int num_samples, int interpolation,
                      const float *rotation, const lrp_post *post, int device, hipStream_t stream, int n_batch = 0,
                      int row_first = 0, int row_count = 0) {
  if (num_samples <= 0) return LRP_OK; // reference loop body never runs: output untouched
  lrp::KParams P = make_params(in, out, num_samples, rotation, post);
  const bool band = row_count > 0 && !(row_first == 0 && row_count == out->height);
  if (band) {
    P.y_offset = row_first;
    P.y_end = row_first + row_count;
  }
  const int oi = out_lens_index(out->lens.type);
  const int im = in_lens_mode(in->lens);
  hipError_t e;
  lrp::TableLease lease; // pins the cached tables until every launch of this call is enqueued (scope end)
  lrp::GeoUse geo;       // this launch's use of the geometry cache (none unless set below)
  const lrp::PlanSwitches sw = plan_switches();
  const lrp::PlanRequest req = plan_request(in, out, num_samples, interpolation, rotation, n_batch, band);
  lrp::PlanFamily family = lrp::plan_family(req, sw);
  lrp::TableFacts tables;
  if (family.wants_tables) {
    // separable output-lens terms (cached per device / lens / size / num_samples)
    const int out_kind = out->lens.type == LRP_RECTILINEAR ? lrp::kRect : lrp::kEquirect;
    e = lrp::get_output_tables(device, out_kind, P.out_lens, out->width, out->height, num_samples, stream, lease,
                               &P.col_tab, &P.row_tab, &tables.plain, &tables.symmetry);
    if (e == hipErrorOutOfMemory) { // the geometry cache of THIS GPU holds what it holds for speed only: give it back, once
      (void)hipGetLastError();
      lrp::geo_release_device(device);
      e = lrp::get_output_tables(device, out_kind, P.out_lens, out->width, out->height, num_samples, stream, lease,
                                 &P.col_tab, &P.row_tab, &tables.plain, &tables.symmetry);
    }
    if (e == hipErrorOutOfMemory) {
      (void)hipGetLastError();
      family.tile = false; // no memory for the tables: per-pixel kernel
    } else if (e != hipSuccess) {
      return hip_fail(e, "output-lens table build");
    } else {
      tables.built = true;
      const lrp::PlanRotation pr = lrp::plan_rotation(req, sw, family, tables);
      P.has_rot = pr.has_rot ? 1 : 0;
      if (pr.wants_xsep)
        P.xsep_tab = lrp::get_xsep_table(device, P.col_tab, out_kind, out->width, num_samples, P.in_lens, im, in->width,
                                         P.in_lon_span, P.has_rot ? P.rot : nullptr, stream, lease);
    }
  }
  if (family.tile) {
    lrp::PlanRotation pr;
    pr.has_rot = P.has_rot != 0;
    const lrp::PlanSharing sh = lrp::plan_sharing(req, sw, family, tables, pr, P.xsep_tab != nullptr);
    const bool window = sh.window;
    P.quad = sh.quad;
    P.win_coef = sh.win_coef, P.win_edge = sh.win_edge, P.win_split = sh.win_split, P.win_tapdma = sh.win_tapdma;
    P.win_mode = sh.win_mode;
    P.alias_pairs = sh.alias_pairs;
    P.frames_per_wave = sh.frames_per_wave;
    if (sh.wants_geo) {
      lrp::GeoKey key;
      std::memset(&key, 0, sizeof(key));
      key.device = device;
      key.out_type = oi == 0 ? lrp::kRect : (oi == 1 ? lrp::kEquidistant : lrp::kEquirect);
      key.in_mode = im;
      key.out_w = out->width, key.out_h = out->height, key.in_w = in->width, key.in_h = in->height;
      key.has_rot = P.has_rot;
      key.num_samples = num_samples;
      key.out_lens = lrp::geo_canonical_lens(P.out_lens, out->lens.type), key.in_lens = lrp::geo_canonical_lens(P.in_lens, in->lens.type);
      if (P.has_rot) std::memcpy(key.rot, P.rot, sizeof(key.rot));
      lrp::geo_acquire(key, sh.geo_want_boxes, stream, &geo);
      lrp::GeoFacts facts;
      facts.mode = geo.mode, facts.lists = geo.lists;
      facts.n_work = geo.n_work, facts.n_runs = geo.n_runs, facts.n_corner_blocks = geo.n_corner_blocks, facts.n_blocks = geo.n_blocks;
      facts.n_wide = geo.n_wide, facts.n_inview = geo.n_inview;
      const lrp::PlanGeo pg = lrp::plan_geo(req, sw, sh, facts);
      if (pg.geo_mode != 0) {
        P.geo_mode = pg.geo_mode;
        P.geo_xy = geo.xy;
        P.geo_box = geo.box;
        P.win_mode = pg.win_mode;
        P.quad = pg.quad;
        P.blocks_per_wave = pg.blocks_per_wave;
        P.rgbaz_runs = pg.rgbaz_runs;
        P.big_windows = pg.big_windows;
        if (pg.listed) { // rendering by block class: the lists of the entry (lrp_params.h "Block lists")
          const uint8_t *const lists = reinterpret_cast<const uint8_t *>(geo.box) + lrp::geo_lists_offset(out->width, out->height);
          P.geo_work = reinterpret_cast<const int32_t *>(lists + (size_t)lrp::kGeoListHeaderWords * 4);
          P.geo_runs = reinterpret_cast<const uint32_t *>(P.geo_work + 2 * lrp::geo_work_capacity(out->width, out->height));
          P.geo_n_work = geo.n_work;
          P.geo_n_runs = geo.n_runs;
          P.geo_work_rec = pg.list_recs ? reinterpret_cast<const int32_t *>(lists + lrp::geo_work_recs_offset(out->width, out->height)) : nullptr;
          P.geo_fill_stride = pg.fill_stride;
          P.geo_fill_per_wave = pg.fill_per_wave;
        }
      }
    }
    const bool listed = P.geo_work != nullptr;
    auto launch = [&]() {
      if (listed) { // the corner runs of this launch's frames, then (or meanwhile, or inside it) everything else
        g_knobs[kKnobListedLaunches].fetch_add(1, std::memory_order_relaxed);
        if (P.geo_fill_per_wave == 0) {
          const hipError_t fe = fill_corner_runs(P, device, stream);
          if (fe != hipSuccess) return fe;
        }
      }
      if (window && P.geo_mode == 2 && P.big_windows != 0) g_knobs[kKnobBigLaunches].fetch_add(1, std::memory_order_relaxed);
      if (window) return lrp::launch_win_bicubic(P, oi, im, stream);
      if (P.geo_mode == 2 && num_samples > 1) return lrp::launch_ss_gather(P, interpolation, im, stream); // (a lane per sub-sample: coalesced loads of the entry)
      if (interpolation == LRP_NEAREST) return lrp::launch_tile_nearest(P, oi, im, stream);
      if (interpolation == LRP_BILINEAR) return lrp::launch_tile_bilinear(P, oi, im, stream);
      return lrp::launch_tile_bicubic(P, oi, im, stream);
    };
    if (n_batch <= 0) {
      e = launch();
    } else {
      e = hipSuccess;
      int first = 0;
      if (P.geo_mode == 1 || P.geo_mode == 3) {
        // the frame that writes the entry goes alone (the instantiations without the frame loop have the side output); the
        // rest of the batch follows on the same stream and reads it
        P.src = in[0].data;
        P.dst = out[0].data;
        e = launch();
        P.geo_mode = 2;
        first = 1;
      }
      for (; first < n_batch && e == hipSuccess; first += lrp::kMaxBatch) {
        P.batch_n = std::min(lrp::kMaxBatch, n_batch - first);
        for (int i = 0; i < P.batch_n; ++i) {
          P.batch_src[i] = in[first + i].data;
          P.batch_dst[i] = out[first + i].data;
        }
        e = launch();
      }
    }
    if (e == hipSuccess && window && (geo.mode == 1 || geo.mode == 3) && geo.host_counts != nullptr) {
      // the launch above wrote the box records and the class bytes of the entry: the block lists (rectilinear source) and the
      // census of its windows are built behind it, and their header follows the records to the host (page-locked; read once
      // the records' event has completed)
      const uint8_t *const header = reinterpret_cast<const uint8_t *>(geo.box) + lrp::geo_lists_offset(out->width, out->height);
      const bool with_lists = im == lrp::kInRect && knob(kKnobGeoLists) != 0, with_census = knob(kKnobGeoCensus) != 0 && im != lrp::kInEquidistant;
      if (!with_lists && !with_census) {
        // (nothing to tell the host about this entry)
      } else if ((!with_lists || lrp::launch_geo_build_lists(geo.box, out->width, out->height, P.alias_pairs, stream) == hipSuccess) &&
          (!with_census || lrp::launch_geo_census(geo.box, out->width, out->height, in->width, in->height, !with_lists, stream) == hipSuccess) &&
          hipMemcpyAsync(geo.host_counts, header, (size_t)lrp::kGeoListHeaderWords * 4, hipMemcpyDeviceToHost, stream) == hipSuccess)
        geo.lists_enqueued = true;
      else
        (void)hipGetLastError(); // (no lists for this entry: every launch enumerates the frame)
    }
    lrp::geo_launched(&geo, stream, e == hipSuccess);
  } else {
    const int n = n_batch > 0 ? n_batch : 1;
    e = hipSuccess;
    // The run-time channel path holds up to 8 channels of a pixel in registers; wider texels (the
    // reference loop is generic in C, src/reproject.cpp:50,76,134) are rendered 8 channels per launch.
    // post_process touches channels 0-2 only (:423-434): they are all in the first group.
    const int C = out->channels, group = C == 4 ? 4 : 8;
    for (int i = 0; i < n && e == hipSuccess; ++i)
      for (int c0 = 0; c0 < C && e == hipSuccess; c0 += group) {
        P.src = in[i].data + c0;
        P.dst = out[i].data + c0;
        P.ch_count = std::min(group, C - c0);
        P.has_post = post != nullptr && c0 == 0;
        if (interpolation == LRP_NEAREST)
          e = lrp::launch_nearest(P, oi, im, stream);
        else if (interpolation == LRP_BILINEAR)
          e = lrp::launch_bilinear(P, oi, im, stream);
        else
          e = lrp::launch_bicubic(P, oi, im, stream);
      }
  }
  if (e != hipSuccess) return hip_fail(e, "reproject kernel launch");
  return LRP_OK;
}

// Grow-only device / pinned buffer.
struct Buffer {
  void *ptr = nullptr;
  size_t cap = 0;
  bool pinned_host = false;
  int reserve(size_t bytes) {
    if (bytes <= cap) return LRP_OK;
    release();
    hipError_t e = pinned_host ? hipHostMalloc(&ptr, bytes, hipHostMallocDefault) : hipMalloc(&ptr, bytes);
    if (e == hipErrorOutOfMemory && !pinned_host) { // the geometry cache of THIS GPU holds what it holds for speed only: give it back, once
      (void)hipGetLastError();
      int dev = -1;
      if (hipGetDevice(&dev) == hipSuccess) lrp::geo_release_device(dev); // (the allocation is on the calling thread's current device)
      e = hipMalloc(&ptr, bytes);
    }
    if (e != hipSuccess) {
      ptr = nullptr;
      cap = 0;
      return hip_fail(e, pinned_host ? "hipHostMalloc" : "hipMalloc");
    }
    cap = bytes;
    return LRP_OK;
  }
  void release() {
    if (ptr) {
      if (pinned_host)
        (void)hipHostFree(ptr);
      else
        (void)hipFree(ptr);
    }
    ptr = nullptr;
    cap = 0;
  }
};

// One image in flight: device source + destination and the events that hand it
// from the upload stream to the compute stream to the download stream.
struct Slot {
  Buffer d_in, d_out;
  Buffer d_in_packed, d_out_packed; // the frame in its file format (lrp_context_submit_packed)
  hipEvent_t uploaded = nullptr, computed = nullptr, downloaded = nullptr;
  bool used = false;
};

} // namespace

// Three-stage pipeline: H2D on `up`, kernels on `run`, D2H on `down`, so that the
// upload of image i+1, the kernel of image i and the download of image i-1 use
// both PCIe directions and the GPU at the same time (a stream per image does not:
// its own H2D -> kernel -> D2H chain keeps one DMA direction idle).
struct lrp_context {
  std::mutex mutex; // submissions from several host threads are serialised
  int device = 0;
  hipStream_t up = nullptr, run = nullptr, down = nullptr;
  // Consecutive images alternate between two compute streams (contexts with more than one slot): the tail of image i's
  // launch — the last 6 % of a single 4K launch run with the wave slots draining — overlaps the head of image i + 1's.
  // Same-box A/B, single launches of one geometry, wall time per launch: headline 115.2 -> 107.7 us, equirect -> rect
  // 109.0 -> 102.0, equirect -> fisheye rotated 146.4 -> 138.1 (three streams: no further gain).
  hipStream_t run2 = nullptr;
  unsigned run_next = 0;
  std::vector<Slot> slots;
  size_t next = 0;
};

extern "C" {

int lrp_abi_version(void) { return LRP_ABI_VERSION; }

int lrp_debug_kernel(int choice) {
  if (choice < 0 || choice > 3) return kernel_choice();
  return g_knobs[kKnobKernel].exchange(choice, std::memory_order_relaxed);
}

int lrp_debug_set(const char *name, int value) {
  if (!name) return -1;
  for (int k = 0; k < kKnobCount; ++k)
    if (std::strcmp(name, kKnobs[k].name) == 0) {
      if (value < kKnobs[k].lo || value > kKnobs[k].hi) return knob(k);
      return g_knobs[k].exchange(value, std::memory_order_relaxed);
    }
  return -1;
}

void lrp_release_cached_tables(void) {
  lrp::geo_release_all();
  lrp::release_output_tables();
}

int lrp_geometry_cache_configure(long long max_bytes, int min_sightings) {
  lrp::geo_configure(max_bytes, min_sightings);
  return LRP_OK;
}

void lrp_geometry_cache_stats(lrp_geometry_cache_info *out) {
  if (!out) return;
  lrp::GeoStats st{};
  lrp::geo_stats(&st);
  out->bytes = st.bytes;
  out->max_bytes = st.max_bytes;
  out->entries = st.entries;
  out->fills = st.fills;
  out->hits = st.hits;
  out->bypasses = st.bypasses;
  out->evictions = st.evictions;
}

int lrp_device_count(void) { return device_count_cached(); }

const char *lrp_strerror(int status) {
  switch (status) {
  case LRP_OK: return "ok";
  case LRP_ERR_OUTPUT_LENS: return "Output lens type not supported.";
  case LRP_ERR_INPUT_LENS: return "Input lens type not supported.";
  case LRP_ERR_INTERPOLATION: return "Interpolation method not supported.";
  case LRP_ERR_CHANNELS: return "channel count mismatch or unsupported";
  case LRP_ERR_BAD_DIMS: return "bad image dimensions";
  case LRP_ERR_NULL: return "null pointer";
  case LRP_ERR_NO_DEVICE: return "no usable HIP device";
  case LRP_ERR_HIP: return "HIP runtime error";
  case LRP_ERR_OOM: return "out of memory";
  case LRP_ERR_BAD_ARG: return "bad argument";
  default: return "unknown status";
  }
}

const char *lrp_last_error(void) { return g_last_error.c_str(); }

int lrp_reproject_device(const lrp_image *in, lrp_image *out, int num_samples, int interpolation,
                         const float *rotation, const lrp_post *post, int device, void *stream) {
  int st = validate(in, out, interpolation, true);
  if (st != LRP_OK) return st;
  st = select_device(device);
  if (st != LRP_OK) return st;
  return enqueue_reproject(in, out, num_samples, interpolation, rotation, post, device, (hipStream_t)stream);
}

int lrp_reproject_rows_device(const lrp_image *in, lrp_image *out, int num_samples, int interpolation, const float *rotation,
                              const lrp_post *post, int row_first, int row_count, int device, void *stream) {
  int st = validate(in, out, interpolation, true);
  if (st != LRP_OK) return st;
  if (row_first < 0 || row_count < 0 || row_first > out->height - row_count)
    return fail(LRP_ERR_BAD_ARG, "row band outside the output image");
  st = select_device(device);
  if (st != LRP_OK) return st;
  if (row_count == 0) return LRP_OK;
  return enqueue_reproject(in, out, num_samples, interpolation, rotation, post, device, (hipStream_t)stream, 0, row_first, row_count);
}

namespace {
// Side streams of lrp_reproject_multi_device, per device, created on first use and kept.
constexpr int kMaxSide = kMaxSideStreams;
struct MultiFork {
  std::mutex busy; // one fork / join being enqueued at a time per device (the events are re-recorded by every call)
  hipStream_t side[kMaxSide] = {};
  hipEvent_t forked = nullptr, joined[kMaxSide] = {};
};
std::mutex g_fork_registry_mutex;
std::map<int, std::unique_ptr<MultiFork>> g_forks;
MultiFork *multi_fork(int device) { // null if the streams / events cannot be created (the caller then stays on one stream)
  std::lock_guard<std::mutex> lock(g_fork_registry_mutex);
  std::unique_ptr<MultiFork> &slot = g_forks[device];
  if (!slot) {
    std::unique_ptr<MultiFork> f(new MultiFork);
    bool ok = hipEventCreateWithFlags(&f->forked, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < kMaxSide && ok; ++k)
      ok = hipStreamCreateWithFlags(&f->side[k], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&f->joined[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      return nullptr;
    }
    slot = std::move(f);
  }
  return slot.get();
}
// Side streams lrp_reproject_multi_device deals its launches over besides the caller's (lrp_debug_set("multi_fork", n);
// 0 keeps every launch on the caller's stream; default 1 — measured best, 588 -> 509 us per 8192^2 -> 6 x 2048^2 cubemap; 2-3: 522, 5: 549).
int multi_fork_lanes() { return knob(kKnobMultiFork); }

} // namespace

int lrp_reproject_multi_device(const lrp_image *in, lrp_image *outs, int n_out, int num_samples,
                               int interpolation, const float *rotations, const lrp_post *post, int device,
                               void *stream) {
  if (n_out < 0 || (n_out > 0 && !outs)) return fail(LRP_ERR_BAD_ARG, "bad output array");
  for (int i = 0; i < n_out; ++i) {
    int st = validate(in, &outs[i], interpolation, true);
    if (st != LRP_OK) return st;
  }
  int st = select_device(device);
  if (st != LRP_OK) return st;
  // The launches are independent (one source, disjoint outputs) and each is short (a 2048^2 cubemap face: 70-150 us, one
  // or two rounds of wavefronts): dealt over the caller's stream and two side streams the tail of one launch overlaps
  // the head of the next.  Fork / join by events; everything stays ordered behind the caller's earlier work and in front
  // of its later work.  (Not while the caller's stream is being captured into a graph, and not for a single output.)
  MultiFork *fork = nullptr;
  std::unique_lock<std::mutex> fork_lock;
  hipStream_t lanes[kMaxSide + 1];
  int n_lanes = 1;
  lanes[0] = (hipStream_t)stream;
  const int want_side = std::min(multi_fork_lanes(), n_out - 1);
  bool forked = false;
  if (want_side > 0) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) (void)hipGetLastError();
    if (cap == hipStreamCaptureStatusNone && (fork = multi_fork(device)) != nullptr) {
      fork_lock = std::unique_lock<std::mutex>(fork->busy);
      bool ok = hipEventRecord(fork->forked, (hipStream_t)stream) == hipSuccess;
      for (int k = 0; k < want_side && ok; ++k) ok = hipStreamWaitEvent(fork->side[k], fork->forked, 0) == hipSuccess;
      if (ok) {
        for (int k = 0; k < want_side; ++k) lanes[n_lanes++] = fork->side[k];
        forked = true;
      } else { // (side streams that already wait on the event just wait for the caller's earlier work: harmless)
        (void)hipGetLastError();
      }
    }
  }
  int result = LRP_OK;
  for (int i = 0; i < n_out && result == LRP_OK; ++i)
    result = enqueue_reproject(in, &outs[i], num_samples, interpolation, rotations ? rotations + 9 * i : nullptr, post, device, lanes[i % n_lanes]);
  if (fork != nullptr && forked) // join, whatever happened: the caller's stream continues behind the side streams
    for (int k = 0; k + 1 < n_lanes; ++k)
      if (hipEventRecord(fork->joined[k], fork->side[k]) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, fork->joined[k], 0) != hipSuccess) {
        const hipError_t e = hipGetLastError();
        (void)hipStreamSynchronize(fork->side[k]);
        if (result == LRP_OK) result = hip_fail(e == hipSuccess ? hipErrorUnknown : e, "join of lrp_reproject_multi_device");
      }
  return result;
}

namespace {
// Per participant of lrp_reproject_multi: a stream and grow-only buffers, kept for the next call.  A participant is a
// (device, occurrence) pair — a device list may name one GPU several times (tests on a one-GPU box do) — and is locked
// for the duration of a job, so jobs on disjoint device sets run concurrently and jobs that share a GPU queue up on it.
struct MultiPeer {
  int device = -1;
  std::mutex busy;
  hipStream_t stream = nullptr;
  hipEvent_t source_ready = nullptr;
  Buffer src, out;
};
std::mutex g_multi_registry_mutex; // guards the map only, never held while a job runs
std::map<std::pair<int, int>, std::unique_ptr<MultiPeer>> g_multi_peers;

MultiPeer *multi_peer(int device, int occurrence) {
  std::lock_guard<std::mutex> lock(g_multi_registry_mutex);
  std::unique_ptr<MultiPeer> &slot = g_multi_peers[{device, occurrence}];
  if (!slot) {
    slot.reset(new MultiPeer);
    slot->device = device;
  }
  return slot.get();
}
} // namespace

int lrp_reproject_multi(const lrp_image *in, lrp_image *outs, int n_out, int num_samples, int interpolation,
                        const float *rotations, const lrp_post *post, const int *devices, int n_devices) {
  if (n_out < 0 || (n_out > 0 && !outs) || !devices || n_devices < 1 || n_devices > 64)
    return fail(LRP_ERR_BAD_ARG, "bad output array or device list");
  for (int i = 0; i < n_out; ++i) {
    int st = validate(in, &outs[i], interpolation, true);
    if (st != LRP_OK) return st;
  }
  for (int d = 0; d < n_devices; ++d) {
    int st = select_device(devices[d]);
    if (st != LRP_OK) return st;
  }
  if (n_out == 0 || num_samples <= 0) return LRP_OK;
  // participants in list order; locked in (device, occurrence) order so that two jobs never wait for each other
  std::vector<MultiPeer *> peers((size_t)n_devices);
  for (int d = 0; d < n_devices; ++d) {
    int occurrence = 0;
    for (int e = 0; e < d; ++e) occurrence += devices[e] == devices[d];
    peers[(size_t)d] = multi_peer(devices[d], occurrence);
  }
  std::vector<MultiPeer *> order(peers);
  std::sort(order.begin(), order.end(), std::less<MultiPeer *>()); // stable addresses (map of unique_ptr): any total order will do
  std::vector<std::unique_lock<std::mutex>> locks;
  for (MultiPeer *p : order) locks.emplace_back(p->busy);

  const size_t in_bytes = image_bytes(*in);
  // With at least as many outputs as participants whole outputs are dealt round-robin (output i -> participant i % n): a
  // whole image keeps the kernels that share work between mirror images and the geometry cache.  With fewer outputs than
  // participants every participant renders band d of every output (rows are independent, src/reproject.cpp:284).  Either
  // way a participant's output buffer holds its pieces back to back.
  const bool whole_outputs = n_out >= n_devices;
  auto band = [&](int i, int d, int &first, int &count) {
    const lrp_image &o = outs[i];
    if (whole_outputs) {
      first = 0;
      count = (i % n_devices == d) ? o.height : 0;
      return;
    }
    first = (int)((long long)o.height * d / n_devices);
    count = (int)((long long)o.height * (d + 1) / n_devices) - first;
  };
  // Everything that can fail after the first asynchronous copy has been enqueued goes through `result` and falls
  // through to the synchronisation of every participant's stream below: no copy out of in->data or into outs[i].data
  // is in flight when this function returns, whatever happened.
  int result = LRP_OK;
  auto hip_ok = [&](hipError_t e, const char *what) {
    if (e != hipSuccess && result == LRP_OK) result = hip_fail(e, what);
    return e == hipSuccess && result == LRP_OK;
  };
  // Where a participant's source lives: the first participant on a GPU holds the copy of that GPU (`holder[d] == d`), later
  // occurrences of the same GPU read it in place.  The holders receive it in a binary tree: holder k (k-th distinct GPU)
  // copies from holder k - 2^floor(log2 k) — after round r, 2^r GPUs hold the source and every one of them feeds another,
  // each over its own xGMI link (seven copies out of one root share that root's links and its HBM read bandwidth: 805 MB x 7).
  std::vector<int> holder((size_t)n_devices), holders;
  for (int d = 0; d < n_devices; ++d) {
    holder[(size_t)d] = d;
    for (int e = 0; e < d; ++e)
      if (devices[e] == devices[d]) {
        holder[(size_t)d] = e;
        break;
      }
    if (holder[(size_t)d] == d) holders.push_back(d);
  }
  for (int d = 0; d < n_devices && result == LRP_OK; ++d) {
    MultiPeer &p = *peers[(size_t)d];
    if (!hip_ok(hipSetDevice(p.device), "hipSetDevice")) break;
    if (!p.stream && !hip_ok(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking), "hipStreamCreateWithFlags")) break;
    if (!p.source_ready && !hip_ok(hipEventCreateWithFlags(&p.source_ready, hipEventDisableTiming), "hipEventCreateWithFlags")) break;
    size_t out_bytes = 0;
    for (int i = 0; i < n_out; ++i) {
      int first, count;
      band(i, d, first, count);
      out_bytes += (size_t)count * (size_t)outs[i].width * (size_t)outs[i].channels * 4u;
    }
    if (holder[(size_t)d] == d) result = p.src.reserve(in_bytes);
    if (result == LRP_OK) result = p.out.reserve(out_bytes ? out_bytes : 4);
  }
  // the source: host -> devices[0] once, then device to device down the tree
  MultiPeer &root = *peers[0];
  if (result == LRP_OK && hip_ok(hipSetDevice(root.device), "hipSetDevice") &&
      hip_ok(hipMemcpyAsync(root.src.ptr, in->data, in_bytes, hipMemcpyHostToDevice, root.stream), "hipMemcpyAsync (source upload)"))
    hip_ok(hipEventRecord(root.source_ready, root.stream), "hipEventRecord");
  for (size_t k = 1; k < holders.size() && result == LRP_OK; ++k) {
    size_t top = 1;
    while (top * 2 <= k) top *= 2;
    MultiPeer &p = *peers[(size_t)holders[k]], &from = *peers[(size_t)holders[k - top]];
    if (!hip_ok(hipSetDevice(p.device), "hipSetDevice")) break;
    int can = 0;
    (void)hipDeviceCanAccessPeer(&can, p.device, from.device);
    if (can) {
      const hipError_t e = hipDeviceEnablePeerAccess(from.device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
      (void)hipGetLastError();
    }
    if (can) {
      if (hip_ok(hipStreamWaitEvent(p.stream, from.source_ready, 0), "hipStreamWaitEvent"))
        hip_ok(hipMemcpyPeerAsync(p.src.ptr, p.device, from.src.ptr, from.device, in_bytes, p.stream), "hipMemcpyPeerAsync");
    } else {
      hip_ok(hipMemcpyAsync(p.src.ptr, in->data, in_bytes, hipMemcpyHostToDevice, p.stream), "hipMemcpyAsync (second upload)");
    }
    if (result == LRP_OK) hip_ok(hipEventRecord(p.source_ready, p.stream), "hipEventRecord");
  }
  // later occurrences of a GPU wait for that GPU's copy
  for (int d = 0; d < n_devices && result == LRP_OK; ++d) {
    if (holder[(size_t)d] == d) continue;
    MultiPeer &p = *peers[(size_t)d];
    if (!hip_ok(hipSetDevice(p.device), "hipSetDevice")) break;
    hip_ok(hipStreamWaitEvent(p.stream, peers[(size_t)holder[(size_t)d]]->source_ready, 0), "hipStreamWaitEvent");
  }
  // bands: render into the participant's buffer at the band's own row offset (the kernels address whole images),
  // download each band to its rows of the host output
  for (int d = 0; d < n_devices && result == LRP_OK; ++d) {
    MultiPeer &p = *peers[(size_t)d];
    if (!hip_ok(hipSetDevice(p.device), "hipSetDevice")) break;
    size_t cursor = 0; // floats into p.out
    for (int i = 0; i < n_out && result == LRP_OK; ++i) {
      int first, count;
      band(i, d, first, count);
      if (count == 0) continue;
      const size_t row_floats = (size_t)outs[i].width * (size_t)outs[i].channels;
      lrp_image din = *in, dout = outs[i];
      din.data = (float *)peers[(size_t)holder[(size_t)d]]->src.ptr;
      // a virtual whole image whose rows [first, first + count) are the buffer's [cursor, ...): only those are written
      dout.data = (float *)p.out.ptr + cursor - (size_t)first * row_floats;
      result = enqueue_reproject(&din, &dout, num_samples, interpolation, rotations ? rotations + 9 * i : nullptr, post, p.device,
                                 p.stream, 0, first, count);
      if (result != LRP_OK) break;
      hip_ok(hipMemcpyAsync(outs[i].data + (size_t)first * row_floats, (float *)p.out.ptr + cursor, (size_t)count * row_floats * 4u,
                            hipMemcpyDeviceToHost, p.stream), "hipMemcpyAsync (band download)");
      cursor += (size_t)count * row_floats;
    }
  }
  // the one exit: every participant's stream drained, then the first error (if any)
  for (MultiPeer *p : peers) {
    if (!p->stream) continue;
    (void)hipSetDevice(p->device);
    const hipError_t e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess && result == LRP_OK) result = hip_fail(e, "hipStreamSynchronize");
  }
  return result;
}

int lrp_reproject_batch_device(const lrp_image *ins, lrp_image *outs, int n, int num_samples, int interpolation,
                               const float *rotation, const lrp_post *post, int device, void *stream) {
  if (n < 0 || (n > 0 && (!ins || !outs))) return fail(LRP_ERR_BAD_ARG, "bad image arrays");
  if (n == 0) return LRP_OK;
  for (int i = 0; i < n; ++i) {
    int st = validate(&ins[i], &outs[i], interpolation, true);
    if (st != LRP_OK) return st;
    const bool same = ins[i].width == ins[0].width && ins[i].height == ins[0].height && ins[i].channels == ins[0].channels &&
                      outs[i].width == outs[0].width && outs[i].height == outs[0].height &&
                      std::memcmp(&ins[i].lens, &ins[0].lens, sizeof(lrp_lens)) == 0 &&
                      std::memcmp(&outs[i].lens, &outs[0].lens, sizeof(lrp_lens)) == 0;
    if (!same) return fail(LRP_ERR_BAD_ARG, "the images of a batch must share sizes, channel count and lenses");
  }
  int st = select_device(device);
  if (st != LRP_OK) return st;
  return enqueue_reproject(ins, outs, num_samples, interpolation, rotation, post, device, (hipStream_t)stream, n);
}

int lrp_post_process_device(lrp_image *img, float exposure, float reinhard, int device, void *stream) {
  if (!img || !img->data) return fail(LRP_ERR_NULL, "null image");
  if (img->width < 1 || img->height < 1 || img->channels < 1) return fail(LRP_ERR_BAD_DIMS, "bad image dimensions");
  if (!image_addressable(*img)) return fail(LRP_ERR_BAD_DIMS, "image too large (more than 2^31 floats)");
  int st = select_device(device);
  if (st != LRP_OK) return st;
  hipError_t e = lrp::launch_post_process(img->data, (uint32_t)img->width * (uint32_t)img->height, img->channels,
                                          exposure, reinhard, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "post_process kernel launch");
  return LRP_OK;
}

int lrp_context_create(lrp_context **ctx, int device, int n_streams) {
  if (!ctx || n_streams < 1 || n_streams > 64) return fail(LRP_ERR_BAD_ARG, "bad context arguments");
  int st = select_device(device);
  if (st != LRP_OK) return st;
  lrp_context *c = new (std::nothrow) lrp_context;
  if (!c) return fail(LRP_ERR_OOM, "host allocation failed");
  c->device = device;
  c->slots.resize((size_t)n_streams);
  hipError_t e = hipStreamCreateWithFlags(&c->up, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->run, hipStreamNonBlocking);
  if (e == hipSuccess && n_streams > 1) e = hipStreamCreateWithFlags(&c->run2, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->down, hipStreamNonBlocking);
  for (auto &s : c->slots) {
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.computed, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.downloaded, hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    lrp_context_destroy(c);
    return hip_fail(e, "stream / event creation");
  }
  *ctx = c;
  return LRP_OK;
}

void lrp_context_destroy(lrp_context *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (hipStream_t st : {ctx->up, ctx->run, ctx->run2, ctx->down})
    if (st) {
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
  for (auto &s : ctx->slots) {
    for (hipEvent_t ev : {s.uploaded, s.computed, s.downloaded})
      if (ev) (void)hipEventDestroy(ev);
    s.d_in.release();
    s.d_out.release();
    s.d_in_packed.release();
    s.d_out_packed.release();
  }
  delete ctx;
}

namespace {
bool format_ok(int f) { return f == LRP_PIXEL_F32 || f == LRP_PIXEL_F16 || f == LRP_PIXEL_U8_GAMMA; }

// One image through the three-stage pipeline.  Formats LRP_PIXEL_F32 with packed channels == image
// channels: the host buffers are the kernels' own layout and are copied straight into / out of the
// slot's float buffers; otherwise the frame is uploaded as it is and converted on the device.
int submit_locked(lrp_context *ctx, const lrp_image *in, int in_format, int in_pch, lrp_image *out, int out_format,
                  int out_pch, unsigned out_fill, int num_samples, int interpolation, const float *rotation,
                  const lrp_post *post, int *ticket) {
  const size_t index = ctx->next;
  Slot &s = ctx->slots[index];
  ctx->next = (ctx->next + 1) % ctx->slots.size();
  if (ticket) *ticket = (int)index;
  const size_t in_px = (size_t)in->width * (size_t)in->height, out_px = (size_t)out->width * (size_t)out->height;
  const size_t in_bytes = image_bytes(*in), out_bytes = image_bytes(*out);
  const bool in_plain = in_format == LRP_PIXEL_F32 && in_pch == in->channels;
  const bool out_plain = out_format == LRP_PIXEL_F32 && out_pch == out->channels;
  const size_t in_packed = in_plain ? 0 : in_px * lrp::pixel_bytes(in_format, in_pch);
  const size_t out_packed = out_plain ? 0 : out_px * lrp::pixel_bytes(out_format, out_pch);
  if (in_bytes > s.d_in.cap || out_bytes > s.d_out.cap || in_packed > s.d_in_packed.cap || out_packed > s.d_out_packed.cap) {
    // growing a buffer frees the old one: the slot's previous image must have drained
    if (s.used) LRP_HIP_TRY(hipEventSynchronize(s.downloaded));
    int st = s.d_in.reserve(in_bytes);
    if (st == LRP_OK) st = s.d_out.reserve(out_bytes);
    if (st == LRP_OK && in_packed) st = s.d_in_packed.reserve(in_packed);
    if (st == LRP_OK && out_packed) st = s.d_out_packed.reserve(out_packed);
    if (st != LRP_OK) return st;
  }
  // upload: the slot's source buffers are free once its previous kernels have run
  if (s.used) LRP_HIP_TRY(hipStreamWaitEvent(ctx->up, s.computed, 0));
  LRP_HIP_TRY(hipMemcpyAsync(in_plain ? s.d_in.ptr : s.d_in_packed.ptr, in->data, in_plain ? in_bytes : in_packed,
                             hipMemcpyHostToDevice, ctx->up));
  LRP_HIP_TRY(hipEventRecord(s.uploaded, ctx->up));
  // kernels: after the upload, and after the previous download has read the destination buffers
  const hipStream_t run = (ctx->run2 != nullptr && knob(kKnobContextStreams) != 0 && (ctx->run_next++ & 1u) != 0) ? ctx->run2 : ctx->run;
  LRP_HIP_TRY(hipStreamWaitEvent(run, s.uploaded, 0));
  if (s.used) LRP_HIP_TRY(hipStreamWaitEvent(run, s.downloaded, 0));
  if (!in_plain) {
    hipError_t e = lrp::launch_decode_pixels(s.d_in_packed.ptr, in_format, in_pch, (float *)s.d_in.ptr, in->channels, in_px,
                                             ctx->device, run);
    if (e != hipSuccess) return hip_fail(e, "pixel decode kernel launch");
  }
  lrp_image din = *in, dout = *out;
  din.data = (float *)s.d_in.ptr;
  dout.data = (float *)s.d_out.ptr;
  int st = enqueue_reproject(&din, &dout, num_samples, interpolation, rotation, post, ctx->device, run);
  if (st != LRP_OK) return st;
  if (!out_plain) {
    hipError_t e = lrp::launch_encode_pixels((const float *)s.d_out.ptr, out->channels, s.d_out_packed.ptr, out_format, out_pch,
                                             out_fill, out_px, ctx->device, run);
    if (e != hipSuccess) return hip_fail(e, "pixel encode kernel launch");
  }
  LRP_HIP_TRY(hipEventRecord(s.computed, run));
  // download
  LRP_HIP_TRY(hipStreamWaitEvent(ctx->down, s.computed, 0));
  LRP_HIP_TRY(hipMemcpyAsync(out->data, out_plain ? s.d_out.ptr : s.d_out_packed.ptr, out_plain ? out_bytes : out_packed,
                             hipMemcpyDeviceToHost, ctx->down));
  LRP_HIP_TRY(hipEventRecord(s.downloaded, ctx->down));
  s.used = true;
  return LRP_OK;
}
} // namespace

int lrp_context_submit_packed(lrp_context *ctx, const lrp_image *in, int in_format, int in_packed_channels, lrp_image *out,
                              int out_format, int out_packed_channels, unsigned out_fill, int num_samples,
                              int interpolation, const float *rotation, const lrp_post *post, int *ticket) {
  if (!ctx) return fail(LRP_ERR_NULL, "null context");
  if (ticket) *ticket = -1;
  int st = validate(in, out, interpolation, true);
  if (st != LRP_OK) return st;
  if (!format_ok(in_format) || !format_ok(out_format) || in_packed_channels < 1 || out_packed_channels < 1 ||
      in_packed_channels > 64 || out_packed_channels > 64)
    return fail(LRP_ERR_BAD_ARG, "bad pixel format or packed channel count");
  st = select_device(ctx->device);
  if (st != LRP_OK) return st;
  if (num_samples <= 0) return LRP_OK; // reference loop body never runs: output untouched
  std::lock_guard<std::mutex> lock(ctx->mutex);
  return submit_locked(ctx, in, in_format, in_packed_channels, out, out_format, out_packed_channels, out_fill, num_samples,
                       interpolation, rotation, post, ticket);
}

int lrp_context_submit(lrp_context *ctx, const lrp_image *in, lrp_image *out, int num_samples, int interpolation,
                       const float *rotation, const lrp_post *post) {
  if (!in || !out) return fail(LRP_ERR_NULL, "null image");
  return lrp_context_submit_packed(ctx, in, LRP_PIXEL_F32, in->channels, out, LRP_PIXEL_F32, out->channels, 0u, num_samples,
                                   interpolation, rotation, post, nullptr);
}

int lrp_context_wait_ticket(lrp_context *ctx, int ticket) {
  if (!ctx) return fail(LRP_ERR_NULL, "null context");
  if (ticket < 0) return LRP_OK; // nothing was enqueued (num_samples <= 0)
  if ((size_t)ticket >= ctx->slots.size()) return fail(LRP_ERR_BAD_ARG, "bad ticket");
  int st = select_device(ctx->device);
  if (st != LRP_OK) return st;
  hipEvent_t ev;
  {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    ev = ctx->slots[(size_t)ticket].downloaded;
  }
  // (a later submission that re-used the slot has re-recorded the event behind this image's download
  // on the same stream: waiting for it waits for this image too)
  LRP_HIP_TRY(hipEventSynchronize(ev));
  return LRP_OK;
}

int lrp_context_wait(lrp_context *ctx) {
  if (!ctx) return fail(LRP_ERR_NULL, "null context");
  int st = select_device(ctx->device);
  if (st != LRP_OK) return st;
  int result = LRP_OK;
  std::lock_guard<std::mutex> lock(ctx->mutex);
  for (hipStream_t stream : {ctx->up, ctx->run, ctx->run2, ctx->down}) {
    if (!stream) continue;
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess && result == LRP_OK) result = hip_fail(e, "hipStreamSynchronize");
  }
  return result;
}

namespace {
// The synchronous host-buffer entry points borrow a single-stream context from
// a per-device pool (one per concurrent caller, reused afterwards), so repeated
// calls keep their device buffers and concurrent pool threads never share one.
// Pooled contexts live until process exit.
std::mutex g_pool_mutex;
std::vector<std::vector<lrp_context *>> g_pool; // [device] -> idle contexts

int borrow_context(int device, lrp_context **out) {
  int st = select_device(device);
  if (st != LRP_OK) return st;
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (g_pool.size() <= (size_t)device) g_pool.resize((size_t)device + 1);
    auto &idle = g_pool[(size_t)device];
    if (!idle.empty()) {
      *out = idle.back();
      idle.pop_back();
      return LRP_OK;
    }
  }
  return lrp_context_create(out, device, 1);
}

void return_context(lrp_context *c) {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  g_pool[(size_t)c->device].push_back(c);
}
} // namespace

int lrp_reproject(const lrp_image *in, lrp_image *out, int num_samples, int interpolation, const float *rotation,
                  const lrp_post *post, int device) {
  int st = validate(in, out, interpolation, true);
  if (st != LRP_OK) return st;
  lrp_context *c = nullptr;
  st = borrow_context(device, &c);
  if (st != LRP_OK) return st;
  st = lrp_context_submit(c, in, out, num_samples, interpolation, rotation, post);
  const int st_wait = lrp_context_wait(c);
  return_context(c);
  return st != LRP_OK ? st : st_wait;
}

int lrp_post_process(lrp_image *img, float exposure, float reinhard, int device) {
  if (!img || !img->data) return fail(LRP_ERR_NULL, "null image");
  if (img->width < 1 || img->height < 1 || img->channels < 1) return fail(LRP_ERR_BAD_DIMS, "bad image dimensions");
  if (!image_addressable(*img)) return fail(LRP_ERR_BAD_DIMS, "image too large (more than 2^31 floats)");
  lrp_context *c = nullptr;
  int st = borrow_context(device, &c);
  if (st != LRP_OK) return st;
  st = [&]() -> int {
    Slot &s = c->slots[0];
    const size_t bytes = image_bytes(*img);
    int r = s.d_out.reserve(bytes);
    if (r != LRP_OK) return r;
    LRP_HIP_TRY(hipMemcpyAsync(s.d_out.ptr, img->data, bytes, hipMemcpyHostToDevice, c->run));
    lrp_image d = *img;
    d.data = (float *)s.d_out.ptr;
    r = lrp_post_process_device(&d, exposure, reinhard, device, c->run);
    if (r != LRP_OK) return r;
    LRP_HIP_TRY(hipMemcpyAsync(img->data, s.d_out.ptr, bytes, hipMemcpyDeviceToHost, c->run));
    LRP_HIP_TRY(hipStreamSynchronize(c->run));
    return LRP_OK;
  }();
  return_context(c);
  return st;
}

int lrp_decode_pixels_device(const void *src, int src_format, int src_channels, float *dst, int dst_channels, size_t n_pixels,
                             int device, void *stream) {
  if (!src || !dst) return fail(LRP_ERR_NULL, "null buffer");
  if (!format_ok(src_format) || src_channels < 1 || dst_channels < 1 || src_channels > 64 || dst_channels > 64)
    return fail(LRP_ERR_BAD_ARG, "bad pixel format or channel count");
  int st = select_device(device);
  if (st != LRP_OK) return st;
  if (n_pixels == 0) return LRP_OK;
  hipError_t e = lrp::launch_decode_pixels(src, src_format, src_channels, dst, dst_channels, n_pixels, device, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "pixel decode kernel launch");
  return LRP_OK;
}

int lrp_encode_pixels_device(const float *src, int src_channels, void *dst, int dst_format, int dst_channels, unsigned fill,
                             size_t n_pixels, int device, void *stream) {
  if (!src || !dst) return fail(LRP_ERR_NULL, "null buffer");
  if (!format_ok(dst_format) || src_channels < 1 || dst_channels < 1 || src_channels > 64 || dst_channels > 64)
    return fail(LRP_ERR_BAD_ARG, "bad pixel format or channel count");
  int st = select_device(device);
  if (st != LRP_OK) return st;
  if (n_pixels == 0) return LRP_OK;
  hipError_t e = lrp::launch_encode_pixels(src, src_channels, dst, dst_format, dst_channels, fill, n_pixels, device, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "pixel encode kernel launch");
  return LRP_OK;
}

void lrp_pixel_tables(float decode[256], float threshold[256]) { lrp::pixel_tables_host(decode, threshold); }

int lrp_host_alloc(void **ptr, size_t bytes) {
  if (!ptr) return fail(LRP_ERR_NULL, "null pointer");
  *ptr = nullptr;
  if (device_count_cached() == 0) return fail(LRP_ERR_NO_DEVICE, "no HIP device visible");
  hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
  return LRP_OK;
}

void lrp_host_free(void *ptr) {
  if (ptr) (void)hipHostFree(ptr);
}

int lrp_synth_fill_device(float *data, int width, int height, int channels, uint32_t seed, int depth_channel,
                          int device, void *stream) {
  if (!data) return fail(LRP_ERR_NULL, "null data");
  if (width < 1 || height < 1 || channels < 1) return fail(LRP_ERR_BAD_DIMS, "bad image dimensions");
  const unsigned long long n = (unsigned long long)width * height * channels;
  if (n > kMaxImageFloats) return fail(LRP_ERR_BAD_DIMS, "image too large (more than 2^31 floats)");
  int st = select_device(device);
  if (st != LRP_OK) return st;
  hipError_t e = lrp::launch_synth_fill(data, (uint32_t)n, channels, seed, depth_channel, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "synth_fill kernel launch");
  return LRP_OK;
}

int lrp_checksum_device(const float *data, size_t n, uint64_t *out, int device, void *stream) {
  if (!data || !out) return fail(LRP_ERR_NULL, "null array");
  if (n >= (1ull << 32)) return fail(LRP_ERR_BAD_DIMS, "buffer too large");
  int st = select_device(device);
  if (st != LRP_OK) return st;
  hipError_t e = lrp::launch_checksum(data, n, reinterpret_cast<unsigned long long *>(out), (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "checksum kernel launch");
  return LRP_OK;
}

int lrp_math_eval_device(int func, const float *a, const float *b, float *out, size_t n, int device, void *stream) {
  if (!a || !out) return fail(LRP_ERR_NULL, "null array");
  if (func < 0 || func > 9) return fail(LRP_ERR_BAD_ARG, "unknown function id");
  int st = select_device(device);
  if (st != LRP_OK) return st;
  if (n == 0) return LRP_OK;
  hipError_t e = lrp::launch_math_eval(func, a, b, out, n, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "math_eval kernel launch");
  return LRP_OK;
}

} // extern "C"
