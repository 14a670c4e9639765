// geocache_driver.cpp — CPU test driver of the geometry cache's host logic (csrc/lrp_geocache.cpp) on a FAKE HIP runtime:
// the handful of runtime calls the cache makes are defined here (no GPU, no libamdhip64), record what they were asked to do
// and let the test decide when "device work" completes.  Scenarios are run by name; a failed check prints and exits 1.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <thread>
#include <string>
#include <vector>

#include "lrp_geocache.h"

// ---- the fake runtime ---------------------------------------------------------------------------------------------------
namespace fake {
std::mutex mu; // (the fake runtime is called from several threads in the stress scenario)
struct Event {
  bool pending = false;
  int records = 0;
};
std::set<void *> live_device, live_host;
std::map<hipEvent_t, Event *> events;
std::vector<std::pair<hipStream_t, hipEvent_t>> waits; // hipStreamWaitEvent calls, in order
int device_syncs = 0, stream_syncs = 0, frees = 0, mallocs = 0, frees_of_busy_buffers = 0, set_devices = 0;
int fail_event_creates_after = -1; // >= 0: that many more hipEventCreateWithFlags calls succeed, then they fail
thread_local int current_device = 0;
size_t bytes_live = 0;
std::map<void *, size_t> sizes;
void complete_all() {
  std::lock_guard<std::mutex> l(mu);
  for (auto &kv : events) kv.second->pending = false;
}
} // namespace fake

extern "C" {
hipError_t hipMalloc(void **p, size_t n) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  *p = std::malloc(n ? n : 1);
  fake::live_device.insert(*p);
  fake::sizes[*p] = n;
  fake::bytes_live += n;
  ++fake::mallocs;
  return hipSuccess;
}
hipError_t hipFree(void *p) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  if (!fake::live_device.count(p)) {
    std::printf("FAIL: hipFree of a pointer that is not live\n");
    std::exit(1);
  }
  fake::live_device.erase(p);
  fake::bytes_live -= fake::sizes[p];
  std::free(p);
  ++fake::frees;
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  *p = std::calloc(1, n);
  fake::live_host.insert(*p);
  return hipSuccess;
}
hipError_t hipHostFree(void *p) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  fake::live_host.erase(p);
  std::free(p);
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  if (fake::fail_event_creates_after == 0) return hipErrorOutOfMemory;
  if (fake::fail_event_creates_after > 0) --fake::fail_event_creates_after;
  auto *ev = new fake::Event;
  *e = reinterpret_cast<hipEvent_t>(ev);
  fake::events[*e] = ev;
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  delete fake::events[e];
  fake::events.erase(e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  fake::events[e]->pending = true;
  fake::events[e]->records++;
  return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  return fake::events[e]->pending ? hipErrorNotReady : hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  fake::events[e]->pending = false;
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  fake::waits.emplace_back(s, e);
  return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *st) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  *st = hipStreamCaptureStatusNone;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  ++fake::stream_syncs;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  ++fake::device_syncs;
  return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipGetDevice(int *d) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  *d = fake::current_device;
  return hipSuccess;
}
hipError_t hipSetDevice(int d) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  fake::current_device = d;
  ++fake::set_devices;
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
  std::lock_guard<std::mutex> fake_lock(fake::mu);
  *total_b = (size_t)288 << 30;
  *free_b = *total_b;
  return hipSuccess;
}
}

// ---- scenarios ------------------------------------------------------------------------------------------------------------
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);          \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static lrp::GeoKey key_of(int device, int w, int h, float rot0) {
  lrp::GeoKey k;
  std::memset(&k, 0, sizeof(k));
  k.device = device;
  k.out_type = lrp::kRect;
  k.in_mode = lrp::kInEquirectLoop;
  k.out_w = w, k.out_h = h, k.in_w = 256, k.in_h = 128;
  k.has_rot = 1;
  k.rot[0] = rot0;
  return k;
}
static hipStream_t S(int i) { return reinterpret_cast<hipStream_t>((uintptr_t)(0x1000 + 16 * i)); }

// a writing launch of a geometry followed by its publication
static lrp::GeoUse fill(const lrp::GeoKey &k, hipStream_t s, bool boxes, bool lists = false) {
  lrp::GeoUse u;
  lrp::geo_acquire(k, boxes, s, &u);
  if (u.mode == 1 || u.mode == 3) {
    if (lists && u.host_counts) {
      const uint32_t counts[8] = {800, 12, 40, 1000, 0, 0, 123, 456}; // (4, 5 unused; 6, 7: the census of the entry's windows)
      std::memcpy(u.host_counts, counts, sizeof(counts)); // (what the device-side copy would deliver)
      u.lists_enqueued = true;
    }
    lrp::GeoUse v = u;
    lrp::geo_launched(&v, s, true);
  }
  return u;
}

static int scenario_fill_read_lists() {
  lrp::geo_configure(64 << 20, 1);
  const lrp::GeoKey k = key_of(0, 640, 480, 1.0f);
  const lrp::GeoUse w = fill(k, S(0), true, true);
  CHECK(w.mode == 1 && w.xy != nullptr && w.box != nullptr && w.host_counts != nullptr);
  CHECK(lrp::geo_peek(k, true));
  // a reader on ANOTHER stream while the writer is in flight: ordered on the device behind the map and the records
  fake::waits.clear();
  lrp::GeoUse r;
  lrp::geo_acquire(k, true, S(1), &r);
  CHECK(r.mode == 2 && r.xy == w.xy && !r.lists); // (the lists are not known before the records' event has completed)
  CHECK(fake::waits.size() == 2 && fake::waits[0].first == S(1) && fake::waits[1].first == S(1));
  lrp::geo_launched(&r, S(1), true);
  fake::complete_all();
  fake::waits.clear();
  lrp::geo_acquire(k, true, S(1), &r);
  CHECK(r.mode == 2 && r.lists && r.n_work == 800 && r.n_runs == 12 && r.n_corner_blocks == 40 && r.n_blocks == 1000);
  CHECK(r.n_wide == 123 && r.n_inview == 456);
  CHECK(fake::waits.empty()); // ready parts need no event
  lrp::geo_launched(&r, S(1), true);
  // a nearest / bilinear launch does not ask for the records
  lrp::geo_acquire(k, false, S(2), &r);
  CHECK(r.mode == 2 && !r.lists);
  lrp::geo_launched(&r, S(2), true);
  lrp::GeoStats st;
  lrp::geo_stats(&st);
  CHECK(st.entries == 1 && st.fills == 1 && st.hits == 3 && st.evictions == 0);
  CHECK(fake::device_syncs == 0);
  return 0;
}

// The regions of an entry (lrp_params.h): map, box records + class bytes, list header, work list, runs, work
// records — disjoint, inside the entry, aligned as their readers need (256 bytes for the records and the header, 16 for the
// 32-byte work records the list builder writes as two 16-byte vectors), for sizes around every rounding boundary.
static int scenario_layout() {
  const int sizes[][2] = {{1, 1}, {15, 17}, {16, 16}, {17, 15}, {72, 67}, {127, 129}, {128, 128}, {640, 480}, {2048, 2048}, {4096, 4096}, {8192, 4096}, {33, 4097}};
  for (const auto &wh : sizes) {
    const int w = wh[0], h = wh[1];
    const lrp::GeoLayout L = lrp::geo_layout(w, h, true);
    CHECK(L.xy_bytes % 256 == 0 && L.xy_bytes >= (size_t)w * h * 8);
    const size_t blocks = (size_t)lrp::geo_block_cols(w) * lrp::geo_block_rows(h);
    CHECK(lrp::geo_class_offset(w, h) == blocks * 32);
    CHECK(L.box_bytes == lrp::geo_lists_offset(w, h) && L.box_bytes % 256 == 0 && L.box_bytes >= blocks * 33);
    const size_t cap = lrp::geo_work_capacity(w, h), runs = lrp::geo_run_capacity(w, h);
    CHECK(cap % lrp::kXcds == 0 && cap >= (size_t)lrp::geo_block_cols(w) * lrp::geo_image_block_rows(h));
    CHECK(runs >= (size_t)lrp::geo_block_cols(w) * lrp::geo_image_block_rows(h) / lrp::kGeoRunBlocks);
    const size_t recs = lrp::geo_work_recs_offset(w, h);
    CHECK(recs == (size_t)lrp::kGeoListHeaderWords * 4 + cap * 8 + runs * 16);
    CHECK(recs % 16 == 0);                // (int4 stores of the list builder, dwordx8 scalar loads of the kernel)
    CHECK(L.list_bytes == recs + cap * 32); // the records are the last region
    CHECK(L.bytes() == L.xy_bytes + L.box_bytes + L.list_bytes);
  }
  const lrp::GeoLayout no_boxes = lrp::geo_layout(640, 480, false);
  CHECK(no_boxes.box_bytes == 0 && no_boxes.list_bytes == 0);
  // an entry of sub-samples (num_samples 2-4): a coordinate pair per sub-sample, the ns^2 of a pixel next to each other in the
  // reference's order, no records; the last element lies inside the map
  for (const auto &wh : sizes)
    for (int ns = 2; ns <= 4; ++ns) {
      const int w = wh[0], h = wh[1], n = ns * ns;
      const lrp::GeoLayout L = lrp::geo_layout(w, h, true, ns);
      CHECK(L.box_bytes == 0 && L.list_bytes == 0 && L.xy_bytes % 256 == 0 && L.xy_bytes >= (size_t)w * h * n * 8);
      CHECK(lrp::geo_ss_map_index(0, 0, w, n, 0) == 0 && lrp::geo_ss_map_index(0, 0, w, n, n - 1) == (uint32_t)(n - 1));
      CHECK(lrp::geo_ss_map_index(1, 0, w, n, 0) == (uint32_t)n && lrp::geo_ss_map_index(0, 1, w, n, 0) == (uint32_t)(w * n));
      CHECK(((size_t)lrp::geo_ss_map_index(w - 1, h - 1, w, n, n - 1) + 1) * 8 <= L.xy_bytes);
    }
  return 0;
}

// Entries of sub-samples live beside plain ones: the key carries num_samples; they hold a map and nothing else.
static int scenario_sub_sample_entries() {
  lrp::geo_configure(64 << 20, 1);
  lrp::GeoKey k1 = key_of(0, 320, 200, 3.0f), k2 = k1, k3 = k1;
  k1.num_samples = 1, k2.num_samples = 2, k3.num_samples = 3;
  const lrp::GeoUse a = fill(k1, S(0), true, true);
  const lrp::GeoUse b = fill(k2, S(0), false);
  const lrp::GeoUse c = fill(k3, S(0), false);
  CHECK(a.mode == 1 && b.mode == 1 && c.mode == 1 && a.xy != b.xy && b.xy != c.xy && b.host_counts == nullptr);
  lrp::GeoStats st;
  lrp::geo_stats(&st);
  CHECK(st.entries == 3);
  CHECK(st.bytes == lrp::geo_layout(320, 200, true).bytes() + lrp::geo_layout(320, 200, true, 2).bytes() + lrp::geo_layout(320, 200, true, 3).bytes());
  fake::complete_all();
  lrp::GeoUse r;
  lrp::geo_acquire(k2, false, S(1), &r); // nearest / bilinear / bicubic with num_samples 2: one entry
  CHECK(r.mode == 2 && r.xy == b.xy && !r.lists);
  lrp::geo_launched(&r, S(1), true);
  lrp::geo_acquire(k1, true, S(1), &r);
  CHECK(r.mode == 2 && r.xy == a.xy && r.lists);
  lrp::geo_launched(&r, S(1), true);
  return 0;
}

static int scenario_map_then_boxes() { // a bilinear launch makes the entry, the first bicubic one adds the records
  lrp::geo_configure(64 << 20, 1);
  const lrp::GeoKey k = key_of(0, 320, 200, 2.0f);
  const lrp::GeoUse a = fill(k, S(0), false);
  CHECK(a.mode == 1 && a.host_counts == nullptr);
  CHECK(lrp::geo_peek(k, false) && !lrp::geo_peek(k, true));
  const lrp::GeoUse b = fill(k, S(0), true, true);
  CHECK(b.mode == 3 && b.host_counts != nullptr && b.xy == a.xy);
  fake::complete_all();
  lrp::GeoUse r;
  lrp::geo_acquire(k, true, S(0), &r);
  CHECK(r.mode == 2 && r.lists);
  lrp::geo_launched(&r, S(0), true);
  return 0;
}

static int scenario_eviction_without_device_sync() {
  const lrp::GeoLayout one = lrp::geo_layout(640, 480, true);
  lrp::geo_configure((long long)(2.5 * one.bytes()), 1);
  fake::waits.clear();
  // three geometries of one size under a cap that holds two: the third takes over the buffer of the first while its launches
  // are still "in flight" — ordered by stream waits on the victim's marks, no hipFree, no device synchronisation
  const lrp::GeoUse a = fill(key_of(0, 640, 480, 1.0f), S(0), true);
  const lrp::GeoUse b = fill(key_of(0, 640, 480, 2.0f), S(1), true);
  const int frees0 = fake::frees, mallocs0 = fake::mallocs;
  const lrp::GeoUse c = fill(key_of(0, 640, 480, 3.0f), S(2), true);
  CHECK(a.mode == 1 && b.mode == 1 && c.mode == 1);
  CHECK(c.xy == a.xy);                                    // the retired buffer of the least recently used entry
  CHECK(fake::frees == frees0 && fake::mallocs == mallocs0); // ... taken over, neither freed nor replaced
  bool waited_on_s2 = false;
  for (auto &w : fake::waits) waited_on_s2 |= w.first == S(2);
  CHECK(waited_on_s2); // the new writer's stream waits for the old launches
  CHECK(fake::device_syncs == 0 && fake::stream_syncs == 0);
  lrp::GeoStats st;
  lrp::geo_stats(&st);
  CHECK(st.evictions == 1 && st.entries == 2 && st.bytes <= st.max_bytes);
  // a geometry of ANOTHER size: the victim's buffer does not fit, it stays retired until its events have completed ...
  const lrp::GeoUse d = fill(key_of(0, 1280, 960, 4.0f), S(3), true); // (larger than the cap allows next to two others)
  lrp::geo_stats(&st);
  CHECK(d.mode == 0 || st.bytes <= st.max_bytes);
  fake::complete_all();
  // ... and is returned to the driver by a later caller that needs room, never while a launch may touch it
  const lrp::GeoUse e = fill(key_of(0, 320, 240, 5.0f), S(0), true);
  CHECK(e.mode == 1);
  CHECK(fake::device_syncs == 0);
  return 0;
}

static int scenario_devices_are_independent() {
  const lrp::GeoLayout one = lrp::geo_layout(640, 480, true);
  lrp::geo_configure((long long)(1.5 * one.bytes()), 1);
  fake::current_device = 0;
  const lrp::GeoUse a0 = fill(key_of(0, 640, 480, 1.0f), S(0), true);
  fake::current_device = 1;
  const lrp::GeoUse a1 = fill(key_of(1, 640, 480, 1.0f), S(1), true);
  CHECK(a0.mode == 1 && a1.mode == 1 && a0.xy != a1.xy);
  // device 1 evicts its own entry; device 0's stays
  const lrp::GeoUse b1 = fill(key_of(1, 640, 480, 2.0f), S(1), true);
  CHECK(b1.mode == 1);
  CHECK(lrp::geo_peek(key_of(0, 640, 480, 1.0f), true) && !lrp::geo_peek(key_of(1, 640, 480, 1.0f), true) && lrp::geo_peek(key_of(1, 640, 480, 2.0f), true));
  CHECK(fake::device_syncs == 0);
  fake::current_device = 0;
  return 0;
}

static int scenario_default_cap_and_release() {
  lrp::geo_configure(-2, 1); // the default: min(4 GiB, 2 % of 288 GB) = 4 GiB, at least two entries of the largest geometry
  const lrp::GeoUse a = fill(key_of(0, 8192, 8192, 1.0f), S(0), true);
  const lrp::GeoUse b = fill(key_of(0, 8192, 8192, 2.0f), S(0), true);
  CHECK(a.mode == 1 && b.mode == 1);
  lrp::GeoStats st;
  lrp::geo_stats(&st);
  CHECK(st.entries == 2 && st.evictions == 0 && st.max_bytes >= ((uint64_t)4 << 30) && st.bytes > 2ull * 8192 * 8192 * 8);
  lrp::geo_release_all(); // waits for the marks (events), frees everything
  lrp::geo_stats(&st);
  CHECK(st.entries == 0 && st.bytes == 0 && fake::live_device.empty() && fake::live_host.empty() && fake::events.empty());
  CHECK(fake::device_syncs == 0);
  lrp::geo_configure(0, 1); // off
  lrp::GeoUse u;
  lrp::geo_acquire(key_of(0, 64, 64, 1.0f), true, S(0), &u);
  CHECK(u.mode == 0 && u.entry == nullptr);
  return 0;
}

static int scenario_failed_launch_and_key() {
  lrp::geo_configure(64 << 20, 1);
  const lrp::GeoKey k = key_of(0, 200, 100, 7.0f);
  lrp::GeoUse u;
  lrp::geo_acquire(k, true, S(0), &u);
  CHECK(u.mode == 1);
  lrp::GeoUse v;
  lrp::geo_acquire(k, true, S(1), &v); // claimed by a launch that is being enqueued right now: no cache for this one
  CHECK(v.mode == 0);
  lrp::geo_launched(&u, S(0), false); // the writing launch failed: the entry goes, nothing dangling
  CHECK(!lrp::geo_peek(k, false));
  // the key ignores what the lens type does not have
  lrp::LensP rect{};
  rect.p[0] = 18.0f, rect.p[1] = 123.0f, rect.p[2] = -7.0f, rect.p[3] = 1e30f;
  const lrp::LensP c = lrp::geo_canonical_lens(rect, /*LRP_RECTILINEAR*/ 0);
  CHECK(c.p[0] == 18.0f && c.p[1] == 0.0f && c.p[2] == 0.0f && c.p[3] == 0.0f);
  lrp::LensP eqr{};
  eqr.p[0] = -1.5f, eqr.p[1] = 1.5f, eqr.p[2] = -3.1f, eqr.p[3] = 3.1f;
  const lrp::LensP ce = lrp::geo_canonical_lens(eqr, /*LRP_EQUIRECTANGULAR*/ 4);
  CHECK(std::memcmp(ce.p, eqr.p, sizeof(eqr.p)) == 0);
  return 0;
}

// Out of memory on ONE GPU (lrp_capi.cpp's fallbacks call geo_release_device): that GPU's entries and retired buffers go back to the
// driver behind their events; the other GPU's entry stays, nobody switches devices, what has been seen is not forgotten.
static int scenario_release_one_device() {
  const lrp::GeoLayout one = lrp::geo_layout(640, 480, true);
  lrp::geo_configure((long long)(1.5 * one.bytes()), 2); // (entries from the second sighting on)
  fake::current_device = 0;
  (void)fill(key_of(0, 640, 480, 1.0f), S(0), true); // first sighting: no entry
  const lrp::GeoUse a0 = fill(key_of(0, 640, 480, 1.0f), S(0), true);
  lrp::GeoUse b0;
  lrp::geo_acquire(key_of(0, 640, 480, 1.0f), true, S(0), &b0);
  { lrp::GeoUse v = b0; lrp::geo_launched(&v, S(0), true); }
  fake::current_device = 1;
  (void)fill(key_of(1, 640, 480, 1.0f), S(1), true);
  const lrp::GeoUse a1 = fill(key_of(1, 640, 480, 1.0f), S(1), true);
  CHECK(a0.mode == 1 && b0.mode == 2 && a1.mode == 1);
  // device 1 meets a geometry of another size: its first entry is evicted, the buffer fits nobody and stays retired (its
  // launches are "in flight")
  const lrp::GeoLayout small = lrp::geo_layout(320, 240, true);
  lrp::geo_configure((long long)(1.1 * one.bytes()), 2);
  (void)fill(key_of(1, 320, 240, 2.0f), S(1), true);
  const lrp::GeoUse s1 = fill(key_of(1, 320, 240, 2.0f), S(1), true);
  CHECK(s1.mode == 1);
  lrp::GeoStats st;
  lrp::geo_stats(&st);
  CHECK(st.entries == 2 && st.evictions == 1 && st.bytes == 2 * one.bytes() + small.bytes()); // (a retired buffer counts until it has gone back)
  const int set0 = fake::set_devices, frees0 = fake::frees;
  lrp::geo_release_device(1);
  CHECK(fake::set_devices == set0 && fake::frees == frees0 + 2);
  CHECK(lrp::geo_peek(key_of(0, 640, 480, 1.0f), true) && !lrp::geo_peek(key_of(1, 320, 240, 2.0f), true));
  lrp::geo_stats(&st);
  CHECK(st.entries == 1 && st.bytes == one.bytes());
  const lrp::GeoUse again = fill(key_of(1, 320, 240, 2.0f), S(1), true); // its sightings are remembered: cached at once
  CHECK(again.mode == 1);
  CHECK(fake::device_syncs == 0);
  fake::current_device = 0;
  return 0;
}

// An entry used from many short-lived streams (a context per job): the marks of streams whose launches have completed are dropped
// when a new stream arrives — events do not pile up, a take-over does not wait for a hundred stale events.
static int scenario_marks_are_pruned() {
  lrp::geo_configure(64 << 20, 1);
  const lrp::GeoKey k = key_of(0, 320, 200, 9.0f);
  (void)fill(k, S(0), true);
  fake::complete_all();
  const size_t events0 = fake::events.size();
  for (int i = 1; i <= 100; ++i) {
    lrp::GeoUse r;
    lrp::geo_acquire(k, true, S(i), &r);
    CHECK(r.mode == 2);
    lrp::geo_launched(&r, S(i), true);
    if (i % 10 == 0) fake::complete_all(); // (every tenth job sees the earlier ones finished)
  }
  CHECK(fake::events.size() <= events0 + 11);
  // marks of launches still in flight are kept: a take-over waits for exactly those
  lrp::GeoUse r;
  lrp::geo_acquire(k, true, S(200), &r);
  lrp::geo_launched(&r, S(200), true);
  lrp::geo_acquire(k, true, S(201), &r);
  lrp::geo_launched(&r, S(201), true);
  const lrp::GeoLayout one = lrp::geo_layout(320, 200, true);
  lrp::geo_configure((long long)(1.2 * one.bytes()), 1);
  fake::waits.clear();
  const lrp::GeoUse t = fill(key_of(0, 320, 200, 10.0f), S(300), true); // evicts k, takes its buffer over
  CHECK(t.mode == 1);
  size_t waits_on_new = 0;
  for (auto &w : fake::waits) waits_on_new += w.first == S(300);
  CHECK(waits_on_new >= 2 && waits_on_new <= 12);
  CHECK(fake::device_syncs == 0);
  return 0;
}

// The writing launch of a new entry fails AND no event can be recorded behind it (event creation fails): the stream is drained by
// hand, and the entry — no valid map, nobody who would ever write one — goes instead of staying behind as a permanent bypass.
static int scenario_failed_writer_without_mark() {
  lrp::geo_configure(64 << 20, 1);
  const lrp::GeoKey k = key_of(0, 200, 100, 11.0f);
  lrp::GeoUse u;
  lrp::geo_acquire(k, true, S(0), &u);
  CHECK(u.mode == 1);
  fake::fail_event_creates_after = 0; // the mark's event cannot be created
  const int syncs0 = fake::stream_syncs;
  lrp::geo_launched(&u, S(0), false);
  fake::fail_event_creates_after = -1;
  CHECK(fake::stream_syncs == syncs0 + 1);
  lrp::GeoStats st;
  lrp::geo_stats(&st);
  CHECK(st.entries == 0 && !lrp::geo_peek(k, false));
  lrp::geo_acquire(k, true, S(0), &u); // ... and the next launch of the geometry writes a fresh entry
  CHECK(u.mode == 1);
  lrp::geo_launched(&u, S(0), true);
  CHECK(lrp::geo_peek(k, true));
  // the mark fails but the launch itself was fine: the entry is published and stays
  const lrp::GeoKey k2 = key_of(0, 200, 100, 12.0f);
  lrp::geo_acquire(k2, true, S(0), &u);
  CHECK(u.mode == 1);
  fake::fail_event_creates_after = 0;
  lrp::geo_launched(&u, S(0), true);
  fake::fail_event_creates_after = -1;
  CHECK(lrp::geo_peek(k2, true));
  return 0;
}

// eight threads on two devices, a cap that holds a few entries, geometries drawn from a small pool: fills, hits, evictions,
// take-overs and failed launches race; run under -fsanitize=thread / address by tests/test_geocache_host.py
static int scenario_threads() {
  const lrp::GeoLayout one = lrp::geo_layout(160, 120, true);
  lrp::geo_configure((long long)(3.5 * one.bytes()), 1);
  std::vector<std::thread> threads;
  for (int t = 0; t < 8; ++t)
    threads.emplace_back([t]() {
      fake::current_device = t & 1;
      unsigned rng = 12345u + 977u * (unsigned)t;
      for (int i = 0; i < 400; ++i) {
        rng = rng * 1664525u + 1013904223u;
        const lrp::GeoKey k = key_of(t & 1, 160, 120, (float)((rng >> 16) % 7));
        lrp::GeoUse u;
        lrp::geo_acquire(k, ((rng >> 8) & 3) != 0, S(t), &u);
        if (u.mode == 1 || u.mode == 3) u.lists_enqueued = u.host_counts != nullptr;
        lrp::geo_launched(&u, S(t), ((rng >> 4) & 31) != 0);
        if ((i & 15) == 0) fake::complete_all();
        if ((i & 127) == 0) (void)lrp::geo_peek(k, true);
      }
    });
  for (auto &th : threads) th.join();
  lrp::GeoStats st;
  lrp::geo_stats(&st);
  CHECK(st.hits > 0 && st.evictions > 0 && fake::device_syncs == 0);
  fake::complete_all();
  lrp::geo_release_all();
  lrp::geo_stats(&st);
  CHECK(st.entries == 0 && fake::live_device.empty() && fake::events.empty());
  return 0;
}

int main(int argc, char **argv) {
  const std::string name = argc > 1 ? argv[1] : "";
  int rc = 2;
  if (name == "fill_read_lists") rc = scenario_fill_read_lists();
  if (name == "map_then_boxes") rc = scenario_map_then_boxes();
  if (name == "layout") rc = scenario_layout();
  if (name == "sub_sample_entries") rc = scenario_sub_sample_entries();
  if (name == "eviction") rc = scenario_eviction_without_device_sync();
  if (name == "devices") rc = scenario_devices_are_independent();
  if (name == "default_cap") rc = scenario_default_cap_and_release();
  if (name == "failed_launch_and_key") rc = scenario_failed_launch_and_key();
  if (name == "release_one_device") rc = scenario_release_one_device();
  if (name == "marks_are_pruned") rc = scenario_marks_are_pruned();
  if (name == "failed_writer_without_mark") rc = scenario_failed_writer_without_mark();
  if (name == "threads") rc = scenario_threads();
  fake::complete_all();
  lrp::geo_release_all(); // (the cache is process-wide state: handed back so that the leak checker sees what is really lost)
  if (rc == 0 && !(fake::live_device.empty() && fake::live_host.empty() && fake::events.empty())) {
    std::printf("FAIL: %zu device buffers, %zu host buffers, %zu events left after geo_release_all\n", fake::live_device.size(), fake::live_host.size(), fake::events.size());
    rc = 1;
  }
  if (rc == 0) std::printf("ok %s\n", name.c_str());
  if (rc == 2) std::printf("unknown scenario\n");
  return rc;
}
