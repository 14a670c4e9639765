// lrp_geocache.cpp — geometry-keyed coordinate cache in HBM (see lrp_geocache.h).  Host code only.
//
// An entry is one device allocation: the coordinate map ([out_h][out_w] float2) followed by the window kernel's
// per-block extremes.  The two parts are written by kernels as side outputs and have their own life cycle:
//
//   none -> claimed (a launch that will write the part is being enqueued by some thread)
//        -> filling (that launch is in its stream's queue; `event` is recorded behind it)
//        -> ready   (the event has been seen complete; no more event calls for this part)
//
// A launch on another stream than the writer's waits for the writer through the event (hipStreamWaitEvent); a
// launch that meets a `claimed` part runs without the cache (the writer has nothing recorded yet to wait for).
//
// Lifetime: acquire() pins the entry until launched(); only unpinned entries are evicted.  An unpinned entry may
// still be read by launches that are in some stream's queue: its memory is handed to the next geometry without a
// device synchronisation only when every launch that ever touched it went to the very stream the new writer goes to
// (stream order then does the rest); otherwise the device is synchronised first, like the table cache does.
#include "lrp_geocache.h"

#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

namespace lrp {
namespace {

enum PartState : int { kNone = 0, kClaimed, kFilling, kReady };

struct Part {
  PartState state = kNone;
  hipEvent_t event = nullptr; // recorded behind the launch that writes the part
};

struct Entry {
  GeoKey key;
  char *buf = nullptr;
  size_t cap = 0; // bytes allocated
  GeoLayout layout{};
  Part map, box;
  int pins = 0;
  uint64_t last_use = 0, hits = 0;
  std::vector<hipStream_t> streams; // distinct streams whose launches touched the buffer (at most kMaxStreams, then `many`)
  bool many_streams = false;
};

constexpr size_t kMaxStreams = 4;
constexpr size_t kMaxSightings = 64;
constexpr int kUselessEvictionsBeforeCaution = 4;

std::mutex g_mutex;
std::vector<std::unique_ptr<Entry>> g_entries;
size_t g_max_bytes = (size_t)1 << 30; // per device
int g_min_sightings = 1;
uint64_t g_tick = 0;
int g_useless_evictions = 0; // entries evicted in a row that no launch ever read
struct Sighting {
  GeoKey key;
  int count;
};
std::vector<Sighting> g_sightings;
GeoStats g_stats{};

bool same_key(const GeoKey &a, const GeoKey &b) { return std::memcmp(&a, &b, sizeof(GeoKey)) == 0; }

size_t device_bytes(int device) {
  size_t n = 0;
  for (const auto &e : g_entries)
    if (e->key.device == device) n += e->cap;
  return n;
}

void note_stream(Entry &e, hipStream_t s) {
  if (e.many_streams) return;
  for (hipStream_t t : e.streams)
    if (t == s) return;
  if (e.streams.size() >= kMaxStreams)
    e.many_streams = true;
  else
    e.streams.push_back(s);
}

bool only_stream(const Entry &e, hipStream_t s) {
  if (e.many_streams) return false;
  for (hipStream_t t : e.streams)
    if (t != s) return false;
  return true;
}

void destroy_events(Entry &e) {
  for (Part *p : {&e.map, &e.box})
    if (p->event) {
      (void)hipEventDestroy(p->event);
      p->event = nullptr;
    }
}

// A filling part whose event has completed becomes ready.  False: the part cannot be read (none / claimed).
bool part_usable(Part &p) {
  if (p.state == kFilling && hipEventQuery(p.event) == hipSuccess) p.state = kReady;
  (void)hipGetLastError(); // (hipErrorNotReady is not an error)
  return p.state == kFilling || p.state == kReady;
}

// Frees the entry's memory (device synchronised unless `no_sync`) and removes it.  g_mutex held, entry unpinned.
void drop_entry(size_t index, bool no_sync) {
  Entry &e = *g_entries[index];
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != e.key.device) (void)hipSetDevice(e.key.device);
  if (!no_sync) (void)hipDeviceSynchronize();
  destroy_events(e);
  if (e.buf) (void)hipFree(e.buf);
  if (cur != e.key.device) (void)hipSetDevice(cur);
  if (e.hits == 0)
    ++g_useless_evictions;
  g_stats.evictions++;
  g_entries.erase(g_entries.begin() + (long)index);
}

int sightings_of(const GeoKey &key) { // counts this sighting
  for (auto &s : g_sightings)
    if (same_key(s.key, key)) return ++s.count;
  if (g_sightings.size() >= kMaxSightings) g_sightings.erase(g_sightings.begin());
  g_sightings.push_back(Sighting{key, 1});
  return 1;
}

} // namespace

void geo_acquire(const GeoKey &key, bool want_boxes, hipStream_t stream, GeoUse *use) {
  *use = GeoUse{};
  std::lock_guard<std::mutex> lock(g_mutex);
  auto bypass = [&]() { g_stats.bypasses++; };
  if (g_max_bytes == 0) return bypass();
  // a capturing stream takes neither events of other streams nor a pointer that a later eviction would leave dangling
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
  if (cap != hipStreamCaptureStatusNone) return bypass();

  Entry *found = nullptr;
  for (const auto &e : g_entries)
    if (same_key(e->key, key)) found = e.get();
  if (found) {
    Entry &e = *found;
    if (!part_usable(e.map)) return bypass(); // (claimed by a launch that is being enqueued right now)
    bool write_boxes = false;
    if (want_boxes) {
      if (e.box.state == kNone)
        write_boxes = true; // a nearest / bilinear launch made the entry: this launch adds the extremes
      else if (!part_usable(e.box))
        return bypass();
    }
    // readers on another stream than the writer's wait for it
    for (Part *p : {&e.map, &e.box})
      if (p->state == kFilling && (p == &e.map || (want_boxes && !write_boxes)))
        if (hipStreamWaitEvent(stream, p->event, 0) != hipSuccess) {
          (void)hipGetLastError();
          return bypass();
        }
    if (write_boxes) {
      if (!e.box.event && hipEventCreateWithFlags(&e.box.event, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        e.box.event = nullptr;
        return bypass();
      }
      e.box.state = kClaimed;
      use->mode = 3;
      g_stats.fills++;
    } else {
      use->mode = 2;
      e.hits++;
      g_stats.hits++;
      g_useless_evictions = 0;
    }
    e.pins++;
    e.last_use = ++g_tick;
    note_stream(e, stream);
    use->xy = reinterpret_cast<float *>(e.buf);
    use->box = reinterpret_cast<int32_t *>(e.buf + e.layout.xy_bytes);
    use->entry = &e;
    return;
  }

  // a new geometry
  const int need_sightings = std::max(g_min_sightings, g_useless_evictions >= kUselessEvictionsBeforeCaution ? 2 : 1);
  if (sightings_of(key) < need_sightings) return bypass();
  const GeoLayout layout = geo_layout(key.out_w, key.out_h, true);
  const size_t need = layout.bytes();
  if (need > g_max_bytes) return bypass();
  char *buf = nullptr;
  size_t cap_bytes = 0;
  // make room: least recently used unpinned entries of this device go; the last victim's memory is taken over when it
  // is large enough (no hipFree / hipMalloc, and no synchronisation when all its launches went to this very stream)
  while (device_bytes(key.device) + need > g_max_bytes) {
    long victim = -1;
    for (size_t i = 0; i < g_entries.size(); ++i) {
      const Entry &e = *g_entries[i];
      if (e.key.device != key.device || e.pins != 0 || e.map.state == kClaimed || e.box.state == kClaimed) continue;
      if (victim < 0 || e.last_use < g_entries[(size_t)victim]->last_use) victim = (long)i;
    }
    if (victim < 0) return bypass(); // everything is pinned
    Entry &v = *g_entries[(size_t)victim];
    const bool ordered = only_stream(v, stream);
    if (v.cap >= need && v.cap <= 2 * need && device_bytes(key.device) <= g_max_bytes) {
      // take the buffer over (the device's total does not grow)
      if (!ordered) (void)hipDeviceSynchronize();
      buf = v.buf;
      cap_bytes = v.cap;
      v.buf = nullptr;
      v.cap = 0;
      drop_entry((size_t)victim, true);
      break;
    }
    drop_entry((size_t)victim, false);
  }
  if (!buf) {
    if (hipMalloc(reinterpret_cast<void **>(&buf), need) != hipSuccess) {
      (void)hipGetLastError();
      return bypass();
    }
    cap_bytes = need;
  }
  std::unique_ptr<Entry> e(new Entry);
  e->key = key;
  e->buf = buf;
  e->cap = cap_bytes;
  e->layout = layout;
  bool ok = hipEventCreateWithFlags(&e->map.event, hipEventDisableTiming) == hipSuccess;
  if (ok && want_boxes) ok = hipEventCreateWithFlags(&e->box.event, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    destroy_events(*e);
    (void)hipFree(buf);
    return bypass();
  }
  e->map.state = kClaimed;
  if (want_boxes) e->box.state = kClaimed;
  e->pins = 1;
  e->last_use = ++g_tick;
  note_stream(*e, stream);
  use->mode = 1;
  use->xy = reinterpret_cast<float *>(e->buf);
  use->box = reinterpret_cast<int32_t *>(e->buf + layout.xy_bytes);
  use->entry = e.get();
  g_stats.fills++;
  g_entries.push_back(std::move(e));
}

void geo_launched(GeoUse *use, hipStream_t stream, bool ok) {
  if (!use->entry) return;
  std::lock_guard<std::mutex> lock(g_mutex);
  Entry *e = static_cast<Entry *>(use->entry);
  use->entry = nullptr;
  e->pins--;
  if (use->mode == 2) return;
  // the parts this launch was to write: published behind it, or given up
  const bool wrote_map = use->mode == 1, wrote_box = e->box.state == kClaimed;
  bool published = ok;
  if (ok && wrote_map) published = hipEventRecord(e->map.event, stream) == hipSuccess;
  if (published && wrote_box) published = hipEventRecord(e->box.event, stream) == hipSuccess;
  if (published) {
    if (wrote_map) e->map.state = kFilling;
    if (wrote_box) e->box.state = kFilling;
    return;
  }
  (void)hipGetLastError();
  if (wrote_box) e->box.state = kNone;
  if (wrote_map) { // nothing valid in it: remove the entry (the launch may be running all the same: synchronise)
    e->map.state = kNone;
    for (size_t i = 0; i < g_entries.size(); ++i)
      if (g_entries[i].get() == e && e->pins == 0) {
        drop_entry(i, false);
        break;
      }
  }
}

void geo_configure(long long max_bytes, int min_sightings) {
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (max_bytes >= 0) g_max_bytes = (size_t)max_bytes;
    if (min_sightings >= 1) g_min_sightings = min_sightings;
  }
  if (max_bytes == 0) geo_release_all();
}

void geo_stats(GeoStats *out) {
  std::lock_guard<std::mutex> lock(g_mutex);
  *out = g_stats;
  out->max_bytes = g_max_bytes;
  out->entries = g_entries.size();
  out->bytes = 0;
  for (const auto &e : g_entries) out->bytes += e->cap;
}

void geo_release_all() {
  std::lock_guard<std::mutex> lock(g_mutex);
  for (size_t i = 0; i < g_entries.size();) {
    const Entry &e = *g_entries[i];
    if (e.pins == 0 && e.map.state != kClaimed && e.box.state != kClaimed)
      drop_entry(i, false);
    else
      ++i;
  }
  g_sightings.clear();
  g_useless_evictions = 0;
}

} // namespace lrp
