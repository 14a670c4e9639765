// lrp_geocache.cpp — geometry-keyed coordinate cache in HBM (see lrp_geocache.h).  Host code only.
//
// An entry is one device allocation: the coordinate map ([out_h][out_w] float2) followed by the window kernel's
// per-block extremes, class bytes and block lists.  The map and the records are written by kernels as side outputs and
// have their own life cycle:
//
//   none -> claimed (a launch that will write the part is being enqueued by some thread)
//        -> filling (that launch is in its stream's queue; `event` is recorded behind it)
//        -> ready   (the event has been seen complete; no more event calls for this part)
//
// A launch on another stream than the writer's waits for the writer through the event (hipStreamWaitEvent); a
// launch that meets a `claimed` part runs without the cache (the writer has nothing recorded yet to wait for).
//
// State is PER DEVICE (one mutex, one entry list, one pool each): the launch threads of different GPUs never meet.
//
// Lifetime: acquire() pins the entry until launched(); only unpinned entries are evicted.  Every launch that touches an
// entry leaves a MARK — an event recorded behind it on its stream, one per (entry, stream), re-recorded by the next launch
// of that stream.  An evicted buffer keeps its marks: the stream of the next writer that takes the buffer over waits for
// them on the device (hipStreamWaitEvent), whatever streams they were recorded on, and a buffer nobody takes over is
// returned to the driver only after its marks have completed — by the thread that makes room for a NEW geometry, outside
// the lock.  That hipFree is the one call of a launch path that may stall the other streams of its GPU (the runtime
// synchronises the device for it), and it only happens when a new geometry arrives while the cache is over its cap AND no
// retired buffer fits: geometries of one size rotating through a full cache take buffers over and never free; hits never
// free.  Nothing else on the launch path synchronises a device, and no decision depends on the identity of a stream handle.
// Retired buffers count towards geo_stats().bytes until they have gone back to the driver.
#include "lrp_geocache.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/lrp.h"

namespace lrp {

LensP geo_canonical_lens(const LensP &lens, int lens_type) {
  LensP c = lens;
  // include/lrp.h lrp_lens (reference src/config.hpp:15-37): rectilinear {focal_length}, equidistant {fov},
  // equirectangular {latitude_min, latitude_max, longitude_min, longitude_max}
  const int used = lens_type == LRP_EQUIRECTANGULAR ? 4 : (lens_type == LRP_FISHEYE_EQUISOLID ? 2 : 1);
  for (int i = used; i < 4; ++i) c.p[i] = 0.0f;
  return c;
}

namespace {

enum PartState : int { kNone = 0, kClaimed, kFilling, kReady };

struct Part {
  PartState state = kNone;
  hipEvent_t event = nullptr; // recorded behind the launch that writes the part
};

struct Mark {
  hipStream_t stream;
  hipEvent_t event; // behind the last launch of `stream` that touched the buffer
};

struct Entry {
  GeoKey key;
  char *buf = nullptr;
  size_t cap = 0; // bytes allocated
  GeoLayout layout{};
  Part map, box;
  int pins = 0;
  uint64_t last_use = 0, hits = 0;
  std::vector<Mark> marks;
  // block lists: the header's copy arrives in host_counts behind the launch that builds them
  uint32_t *host_counts = nullptr; // page-locked, kGeoListHeaderWords words
  bool lists_pending = false, lists_known = false;
  uint32_t counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// An evicted buffer: reusable by a stream that waits for `events`, returnable to the driver once they have completed.
struct Retired {
  char *buf = nullptr;
  size_t cap = 0;
  std::vector<hipEvent_t> events;
  uint32_t *host_counts = nullptr; // (a copy into it may be in flight, in front of the events)
};

constexpr int kMaxDevices = 64;
constexpr size_t kMaxSightings = 64;
constexpr int kUselessEvictionsBeforeCaution = 4;
constexpr size_t kDefaultCapCeiling = (size_t)4 << 30;

struct Sighting {
  GeoKey key;
  int count;
};

struct DeviceCache {
  std::mutex mutex;
  std::vector<std::unique_ptr<Entry>> entries;
  std::vector<Retired> retired;
  uint64_t tick = 0;
  int useless_evictions = 0; // entries evicted in a row that no launch ever read
  std::vector<Sighting> sightings;
  GeoStats stats{};
  size_t default_cap = 0;   // min(4 GiB, 2 % of the device's memory), asked once
  size_t largest_need = 0;  // bytes of the largest geometry this device has been asked to hold
};

DeviceCache g_devices[kMaxDevices];
std::atomic<long long> g_max_bytes{-1}; // per device; -1: the default cap
std::atomic<int> g_min_sightings{1};

bool same_key(const GeoKey &a, const GeoKey &b) { return std::memcmp(&a, &b, sizeof(GeoKey)) == 0; }

size_t live_bytes(const DeviceCache &d) {
  size_t n = 0;
  for (const auto &e : d.entries) n += e->cap;
  return n;
}
size_t retired_bytes(const DeviceCache &d) {
  size_t n = 0;
  for (const auto &r : d.retired) n += r.cap;
  return n;
}

// The cap of this device (the current device of the calling thread is `d`'s).
size_t cap_of(DeviceCache &d) {
  const long long set = g_max_bytes.load(std::memory_order_relaxed);
  if (set >= 0) return (size_t)set;
  if (d.default_cap == 0) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
      (void)hipGetLastError();
      total_b = (size_t)64 << 30;
    }
    d.default_cap = std::max<size_t>(std::min<size_t>(kDefaultCapCeiling, total_b / 50), (size_t)64 << 20);
  }
  return std::max(d.default_cap, 2 * d.largest_need);
}

void destroy_part_events(Entry &e) {
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

// The list header arrived with the records: once they are ready, so is its copy on the host.
void refresh_lists(Entry &e) {
  if (e.lists_pending && e.box.state == kReady && e.host_counts) {
    std::memcpy(e.counts, e.host_counts, sizeof(e.counts));
    e.lists_pending = false;
    e.lists_known = true;
  }
}

// Takes the entry out of the cache; its buffer, with every event a later user has to be ordered behind, becomes a Retired.
// d.mutex held, entry unpinned.
Retired retire_entry(DeviceCache &d, size_t index) {
  Entry &e = *d.entries[index];
  Retired r;
  r.buf = e.buf;
  r.cap = e.cap;
  for (const Mark &m : e.marks) r.events.push_back(m.event);
  // (the writers' launches are marked as well; the part events are no longer needed)
  destroy_part_events(e);
  r.host_counts = e.host_counts;
  if (e.hits == 0) ++d.useless_evictions;
  d.stats.evictions++;
  d.entries.erase(d.entries.begin() + (long)index);
  return r;
}

bool events_complete(const Retired &r) {
  for (hipEvent_t ev : r.events)
    if (hipEventQuery(ev) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
  return true;
}

void free_retired(Retired &r, bool wait) { // outside any lock
  for (hipEvent_t ev : r.events) {
    if (wait) (void)hipEventSynchronize(ev);
    (void)hipEventDestroy(ev);
  }
  r.events.clear();
  if (r.buf) (void)hipFree(r.buf);
  if (r.host_counts) (void)hipHostFree(r.host_counts);
  r.buf = nullptr;
  r.host_counts = nullptr;
  (void)hipGetLastError();
}

int sightings_of(DeviceCache &d, const GeoKey &key) { // counts this sighting
  for (auto &s : d.sightings)
    if (same_key(s.key, key)) return ++s.count;
  if (d.sightings.size() >= kMaxSightings) d.sightings.erase(d.sightings.begin());
  d.sightings.push_back(Sighting{key, 1});
  return 1;
}

DeviceCache *cache_of(int device) { return (device >= 0 && device < kMaxDevices) ? &g_devices[device] : nullptr; }

} // namespace

void geo_acquire(const GeoKey &key, bool want_boxes, hipStream_t stream, GeoUse *use) {
  *use = GeoUse{};
  DeviceCache *const dc = cache_of(key.device);
  if (!dc) return;
  DeviceCache &d = *dc;
  std::vector<Retired> to_free; // returned to the driver after the lock is dropped
  {
    std::lock_guard<std::mutex> lock(d.mutex);
    auto bypass = [&]() { d.stats.bypasses++; };
    if (g_max_bytes.load(std::memory_order_relaxed) == 0) return bypass();
    // a capturing stream takes neither events of other streams nor a pointer that a later eviction would leave dangling
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
    if (cap != hipStreamCaptureStatusNone) return bypass();

    Entry *found = nullptr;
    for (const auto &e : d.entries)
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
        if (!e.host_counts && hipHostMalloc(reinterpret_cast<void **>(&e.host_counts), kGeoListHeaderWords * 4, hipHostMallocDefault) != hipSuccess) {
          (void)hipGetLastError();
          e.host_counts = nullptr; // (no lists for this entry)
        }
        e.box.state = kClaimed;
        use->mode = 3;
        use->host_counts = e.host_counts;
        d.stats.fills++;
      } else {
        use->mode = 2;
        e.hits++;
        d.stats.hits++;
        d.useless_evictions = 0;
        if (want_boxes) {
          refresh_lists(e);
          if (e.lists_known) {
            use->lists = true;
            use->n_work = e.counts[0], use->n_runs = e.counts[1], use->n_corner_blocks = e.counts[2], use->n_blocks = e.counts[3], use->n_wide = e.counts[6], use->n_inview = e.counts[7];
          }
        }
      }
      e.pins++;
      e.last_use = ++d.tick;
      use->xy = reinterpret_cast<float *>(e.buf);
      use->box = reinterpret_cast<int32_t *>(e.buf + e.layout.xy_bytes);
      use->entry = &e;
      return;
    }

    // a new geometry
    const int need_sightings = std::max(g_min_sightings.load(std::memory_order_relaxed), d.useless_evictions >= kUselessEvictionsBeforeCaution ? 2 : 1);
    if (sightings_of(d, key) < need_sightings) return bypass();
    const GeoLayout layout = geo_layout(key.out_w, key.out_h, true, key.num_samples > 1 ? key.num_samples : 1);
    const size_t need = layout.bytes();
    if (g_max_bytes.load(std::memory_order_relaxed) < 0) d.largest_need = std::max(d.largest_need, need);
    const size_t cap_bytes_dev = cap_of(d);
    if (need > cap_bytes_dev) return bypass();
    // make room: least recently used unpinned entries go (their buffers retire with their marks) ...
    while (live_bytes(d) + need > cap_bytes_dev) {
      long victim = -1;
      for (size_t i = 0; i < d.entries.size(); ++i) {
        const Entry &e = *d.entries[i];
        if (e.pins != 0 || e.map.state == kClaimed || e.box.state == kClaimed) continue;
        if (victim < 0 || e.last_use < d.entries[(size_t)victim]->last_use) victim = (long)i;
      }
      if (victim < 0) return bypass(); // everything is pinned
      d.retired.push_back(retire_entry(d, (size_t)victim));
    }
    // ... a retired buffer of a fitting size is taken over: this stream waits, on the device, for every launch that touched it
    char *buf = nullptr;
    size_t buf_cap = 0;
    uint32_t *buf_counts = nullptr;
    for (size_t i = 0; i < d.retired.size() && !buf; ++i) {
      Retired &r = d.retired[i];
      if (r.cap < need || r.cap > 2 * need) continue;
      bool ordered = true;
      for (hipEvent_t ev : r.events)
        if (hipStreamWaitEvent(stream, ev, 0) != hipSuccess) {
          (void)hipGetLastError();
          ordered = false;
        }
      if (!ordered) continue; // (stays retired; freed once its events complete)
      for (hipEvent_t ev : r.events) (void)hipEventDestroy(ev); // (a pending wait keeps what it needs of a destroyed event)
      buf = r.buf;
      buf_cap = r.cap;
      buf_counts = r.host_counts;
      d.retired.erase(d.retired.begin() + (long)i);
    }
    // ... and what no longer fits goes back to the driver, outside the lock, once nothing on the device uses it
    for (size_t i = 0; i < d.retired.size();) {
      if (live_bytes(d) + retired_bytes(d) + (buf ? 0 : need) > cap_bytes_dev && events_complete(d.retired[i])) {
        to_free.push_back(std::move(d.retired[i]));
        d.retired.erase(d.retired.begin() + (long)i);
      } else {
        ++i;
      }
    }
    if (!buf) {
      if (hipMalloc(reinterpret_cast<void **>(&buf), need) != hipSuccess) {
        (void)hipGetLastError();
        bypass();
        buf = nullptr;
      }
      buf_cap = need;
    }
    if (buf) {
      std::unique_ptr<Entry> e(new Entry);
      e->key = key;
      e->buf = buf;
      e->cap = buf_cap;
      e->layout = layout;
      e->host_counts = buf_counts;
      bool ok = hipEventCreateWithFlags(&e->map.event, hipEventDisableTiming) == hipSuccess;
      if (ok && want_boxes) ok = hipEventCreateWithFlags(&e->box.event, hipEventDisableTiming) == hipSuccess;
      if (ok && want_boxes && !e->host_counts && hipHostMalloc(reinterpret_cast<void **>(&e->host_counts), kGeoListHeaderWords * 4, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        e->host_counts = nullptr; // (no lists for this entry)
      }
      if (!ok) {
        (void)hipGetLastError();
        destroy_part_events(*e);
        Retired r; // (nothing was launched on it: free at once)
        r.buf = buf;
        r.cap = buf_cap;
        r.host_counts = e->host_counts;
        to_free.push_back(std::move(r));
        bypass();
      } else {
        e->map.state = kClaimed;
        if (want_boxes) e->box.state = kClaimed;
        e->pins = 1;
        e->last_use = ++d.tick;
        use->mode = 1;
        use->xy = reinterpret_cast<float *>(e->buf);
        use->box = reinterpret_cast<int32_t *>(e->buf + layout.xy_bytes);
        use->entry = e.get();
        use->host_counts = want_boxes ? e->host_counts : nullptr;
        d.stats.fills++;
        d.entries.push_back(std::move(e));
      }
    }
  }
  for (Retired &r : to_free) free_retired(r, false);
}

bool geo_peek(const GeoKey &key, bool want_boxes) {
  DeviceCache *const dc = cache_of(key.device);
  if (!dc || g_max_bytes.load(std::memory_order_relaxed) == 0) return false;
  std::lock_guard<std::mutex> lock(dc->mutex);
  for (const auto &e : dc->entries)
    if (same_key(e->key, key)) return part_usable(e->map) && (!want_boxes || part_usable(e->box));
  return false;
}

void geo_launched(GeoUse *use, hipStream_t stream, bool ok) {
  if (!use->entry) return;
  Entry *e = static_cast<Entry *>(use->entry);
  DeviceCache &d = *cache_of(e->key.device);
  Retired dead; // an entry whose writing launch failed
  bool have_dead = false, marked = false;
  {
    std::lock_guard<std::mutex> lock(d.mutex);
    use->entry = nullptr;
    // the mark of this stream: behind the launch that has just been enqueued
    Mark *mark = nullptr;
    for (Mark &m : e->marks)
      if (m.stream == stream) mark = &m;
    if (!mark) {
      // a new stream: the marks of streams whose launches have completed order nothing any more — dropped here, so that an entry
      // used from short-lived streams (a context per job) does not collect an event per stream that ever touched it
      for (size_t i = 0; i < e->marks.size();) {
        if (hipEventQuery(e->marks[i].event) == hipSuccess) {
          (void)hipEventDestroy(e->marks[i].event);
          e->marks.erase(e->marks.begin() + (long)i);
        } else {
          (void)hipGetLastError(); // (hipErrorNotReady is not an error)
          ++i;
        }
      }
      hipEvent_t ev = nullptr;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
        e->marks.push_back(Mark{stream, ev});
        mark = &e->marks.back();
      } else {
        (void)hipGetLastError();
      }
    }
    marked = mark && hipEventRecord(mark->event, stream) == hipSuccess;
    if (!marked) (void)hipGetLastError();
    if (marked) e->pins--; // (an unmarked launch keeps its pin until the stream has drained, below)
    if (use->mode != 2) {
      // the parts this launch was to write: published behind it, or given up
      const bool wrote_map = use->mode == 1, wrote_box = e->box.state == kClaimed;
      bool published = ok;
      if (ok && wrote_map) published = hipEventRecord(e->map.event, stream) == hipSuccess;
      if (published && wrote_box) published = hipEventRecord(e->box.event, stream) == hipSuccess;
      if (published) {
        if (wrote_map) e->map.state = kFilling;
        if (wrote_box) {
          e->box.state = kFilling;
          e->lists_pending = use->lists_enqueued && e->host_counts != nullptr;
        }
      } else {
        (void)hipGetLastError();
        if (wrote_box) e->box.state = kNone;
        if (wrote_map) { // nothing valid in it: the entry goes (the launch may be running all the same: its buffer waits for the mark)
          e->map.state = kNone;
          if (marked && e->pins == 0)
            for (size_t i = 0; i < d.entries.size(); ++i)
              if (d.entries[i].get() == e) {
                dead = retire_entry(d, i);
                have_dead = true;
                break;
              }
        }
      }
    }
  }
  if (have_dead) free_retired(dead, true);
  if (!marked) {
    // no event could be recorded behind this launch: a later take-over or free of the buffer only sees marks, so the
    // stream is drained by hand before the pin goes
    (void)hipStreamSynchronize(stream);
    (void)hipGetLastError();
    Retired orphan;
    bool have_orphan = false;
    {
      std::lock_guard<std::mutex> lock(d.mutex);
      e->pins--;
      // ... and an entry whose writing launch failed as well has no valid map and nobody who would write one (geo_acquire
      // bypasses a found entry whose map is unusable): it goes now — the stream has drained, nothing is in flight on it
      if (e->pins == 0 && e->map.state == kNone)
        for (size_t i = 0; i < d.entries.size(); ++i)
          if (d.entries[i].get() == e) {
            orphan = retire_entry(d, i);
            have_orphan = true;
            break;
          }
    }
    if (have_orphan) free_retired(orphan, true);
  }
}

void geo_configure(long long max_bytes, int min_sightings) {
  if (max_bytes >= 0 || max_bytes == -2) g_max_bytes.store(max_bytes == -2 ? -1 : max_bytes, std::memory_order_relaxed);
  if (min_sightings >= 1) g_min_sightings.store(min_sightings, std::memory_order_relaxed);
  if (max_bytes == 0) geo_release_all();
}

void geo_stats(GeoStats *out) {
  *out = GeoStats{};
  const long long set = g_max_bytes.load(std::memory_order_relaxed);
  for (DeviceCache &d : g_devices) {
    std::lock_guard<std::mutex> lock(d.mutex);
    out->entries += d.entries.size();
    out->bytes += live_bytes(d) + retired_bytes(d);
    out->fills += d.stats.fills;
    out->hits += d.stats.hits;
    out->bypasses += d.stats.bypasses;
    out->evictions += d.stats.evictions;
    if (set < 0 && d.default_cap != 0 && out->max_bytes == 0) out->max_bytes = std::max(d.default_cap, 2 * d.largest_need);
  }
  if (set >= 0) out->max_bytes = (uint64_t)set;
  else if (out->max_bytes == 0) out->max_bytes = kDefaultCapCeiling;
}

namespace {
// Retires every unpinned entry of `dev` and returns its buffers to the driver behind their events.  The calling thread's
// current device must be `dev` when anything is freed.
void release_device(int dev, bool forget_sightings) {
  DeviceCache &d = g_devices[dev];
  std::vector<Retired> gone;
  {
    std::lock_guard<std::mutex> lock(d.mutex);
    for (size_t i = 0; i < d.entries.size();) {
      const Entry &e = *d.entries[i];
      if (e.pins == 0 && e.map.state != kClaimed && e.box.state != kClaimed)
        gone.push_back(retire_entry(d, i));
      else
        ++i;
    }
    for (Retired &r : d.retired) gone.push_back(std::move(r));
    d.retired.clear();
    if (forget_sightings) {
      d.sightings.clear();
      d.useless_evictions = 0;
    }
  }
  for (Retired &r : gone) free_retired(r, true);
}
bool device_holds_anything(int dev) {
  DeviceCache &d = g_devices[dev];
  std::lock_guard<std::mutex> lock(d.mutex);
  return !d.entries.empty() || !d.retired.empty() || !d.sightings.empty();
}
} // namespace

void geo_release_device(int device) {
  if (cache_of(device)) release_device(device, false); // (out of memory is no reason to forget what has been seen)
}

void geo_release_all() {
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  if (!have_cur) (void)hipGetLastError();
  for (int dev = 0; dev < kMaxDevices; ++dev) {
    if (!device_holds_anything(dev)) continue;
    if (have_cur && cur != dev) (void)hipSetDevice(dev);
    release_device(dev, true);
    if (have_cur && cur != dev) (void)hipSetDevice(cur);
  }
}

} // namespace lrp
