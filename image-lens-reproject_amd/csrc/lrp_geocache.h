// lrp_geocache.h — geometry-keyed coordinate cache in HBM (host side; layout in lrp_params.h).
//
// The reference calls reproject() once per file with one geometry for the whole run
// (src/main.cpp:576-598): lenses, sizes and rotation do not change, only the pixels do.  Everything
// the hot loop derives from the geometry alone (src/reproject.cpp:287-324: pixel -> ray -> rotation
// -> source lens -> texel coordinates) is therefore the same in every call.  A batched launch shares
// it between its frames in registers; single launches share it through this cache: the first launch
// of a geometry writes the coordinates of every output pixel (and the window extremes of every block
// of the bicubic window kernel) as a side output, later launches of the same geometry load them.
// Loaded values are the stored ones, so the rendered bits do not change.
//
// Bounded (bytes per device, least recently used entries go first), opt-out through the C ABI
// (lrp_geometry_cache_configure), freed by lrp_release_cached_tables.  State is per device: a launch
// thread of one GPU never takes a lock, waits for an event or synchronises on behalf of another GPU
// (out of memory included: geo_release_device).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "lrp_params.h"

namespace lrp {

// Everything the coordinates of src/reproject.cpp:323-324 depend on (compared byte for byte; the caller zeroes what the
// lens type does not use, geo_canonical_lens).
struct GeoKey {
  int32_t device;
  int32_t out_type, in_mode; // output lens (kRect / kEquidistant / kEquirect), input mode (kIn*)
  int32_t out_w, out_h, in_w, in_h;
  int32_t has_rot;
  int32_t num_samples; // 1; 2-4: the entry of the supersampling instantiations (a coordinate pair per sub-sample, no records)
  LensP out_lens, in_lens;
  float rot[9];
};
// The lens parameters a geometry depends on: the members of the union that `lens_type` (include/lrp.h numbering) does not
// have are indeterminate in a caller's struct (a C caller sets focal_length and nothing else) and are zeroed in the key.
// Signed zeros and NaN payloads are kept apart: a miss is only slower, a false hit would change bits.
LensP geo_canonical_lens(const LensP &lens, int lens_type);

// What a launch does with the cache: P.geo_mode and the entry's pointers.
struct GeoUse {
  int mode = 0;          // 0: no cache for this launch; 1: write map + boxes; 3: write boxes only; 2: read
  float *xy = nullptr;
  int32_t *box = nullptr;
  void *entry = nullptr; // opaque; pinned until geo_launched
  // Block lists (lrp_params.h): a launch that writes the boxes (mode 1 with boxes, mode 3) also enqueues geo_build_lists
  // and a copy of the list header into `host_counts` (page-locked) and says so in `lists_enqueued`; a reading launch finds
  // `lists` set once that copy has been seen complete, with the counts of the header.
  uint32_t *host_counts = nullptr;
  bool lists_enqueued = false;
  bool lists = false;
  uint32_t n_work = 0, n_runs = 0, n_corner_blocks = 0, n_blocks = 0;
  uint32_t n_wide = 0, n_inview = 0; // the census of the entry (lrp_geo_lists.hip geo_census): blocks in view whole / those of them no 10 KiB window stages
};

// Decides, for a launch that is about to be enqueued on `stream` (device already selected), whether it reads the
// entry of `key`, writes it, or runs without the cache; want_boxes: the launch is the window kernel (needs the
// per-block extremes too).  A reader on another stream than the entry's writer is made to wait for the writer
// (hipStreamWaitEvent).  Never fails: any problem (no memory, a capturing stream, the cache switched off) is mode 0.
void geo_acquire(const GeoKey &key, bool want_boxes, hipStream_t stream, GeoUse *use);
// True: the entry of `key` exists and a geo_acquire now would read it (mode 2).  An observer: it claims nothing and marks nothing
// (the host tests of the cache, tests/native/geocache_driver.cpp, look at the cache through it; no launch path calls it).
bool geo_peek(const GeoKey &key, bool want_boxes);
// After the launch has been enqueued (ok) or has failed to: publishes a written entry, marks the stream, unpins.
void geo_launched(GeoUse *use, hipStream_t stream, bool ok);

struct GeoStats {
  uint64_t bytes, max_bytes, entries, fills, hits, bypasses, evictions;
};
// max_bytes < 0 / min_sightings < 1: keep the current value.  max_bytes == 0 switches the cache off (and frees it).
// max_bytes == -2: back to the default cap (min(4 GiB, 2 % of the device's memory), at least two entries of the largest
// geometry seen on that device).
void geo_configure(long long max_bytes, int min_sightings);
void geo_stats(GeoStats *out);
void geo_release_all(); // every device: waits for the launches that touched the entries (their events), frees everything unpinned
// The same for ONE device (the calling thread's current device is `device`): what a launch path of that GPU may call when it
// runs out of memory — it neither waits for, nor frees, nor forgets anything of another GPU.
void geo_release_device(int device);

} // namespace lrp
