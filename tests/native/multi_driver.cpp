// multi_driver.cpp — CPU test driver of lrp_reproject_multi (csrc/lrp_capi.cpp: one source, several outputs, several GPUs) on a
// FAKE HIP runtime with 2, 3 and 8 DISTINCT devices.  The real host code of the library is linked in (lrp_capi.cpp, lrp_plan.cpp,
// lrp_geocache.cpp, lrp_host_util.cpp); the runtime calls it makes and the kernel launchers of the .hip units are defined here and
// only RECORD what they were asked to do.  On the round's GPU boxes this entry point has only ever run with GPU 0 named eight
// times (tests/test_gpu_multi_gpu.py); what is checked here is everything that differs between that and eight physical GPUs:
//   * the source goes host -> device 0 ONCE and then device to device down a binary tree: participant k (k-th distinct GPU) copies
//     from participant k - 2^floor(log2 k), on its OWN stream, behind the event recorded after the copy that filled its parent;
//   * later occurrences of a GPU in the list wait for that GPU's copy and fetch nothing;
//   * every call of the runtime that touches a device's stream / memory runs with that device current;
//   * work split: whole outputs round-robin with at least as many outputs as participants, else row band d of every output on
//     participant d — every output row rendered exactly once and downloaded exactly once, from the participant that rendered it;
//   * EVERY exit path (a failing peer copy, a failing launch, a failing download, out of memory) synchronises every participant's
//     stream before returning: no copy out of the caller's source or into its outputs is in flight afterwards;
//   * participants are locked in one global order: jobs over [0 1 2] and [2 1 0] running at once do not deadlock.
// Scenarios are run by name; a failed check prints and exits 1.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lrp.h"
#include "lrp_params.h"
#include "lrp_tables.h"

// ---- the fake runtime ---------------------------------------------------------------------------------------------------
namespace fake {
std::mutex mu;
int n_devices = 8;
bool peers_reachable = true;
thread_local int current_device = 0;
struct Stream {
  int device;
  int id;
  int syncs = 0;
  int pending = 0; // operations enqueued since the last synchronisation
};
struct Event {
  int recorded_on = -1; // stream id
  int after_op = -1;    // index of the last operation of that stream at the time
};
struct Op { // everything enqueued on a stream, in order
  std::string what; // "h2d", "d2h", "peer", "launch", "wait", "record"
  int device;       // current device of the calling thread
  int stream;       // stream id
  const void *dst = nullptr, *src = nullptr;
  size_t bytes = 0;
  int dst_device = -1, src_device = -1; // peer copies
  int event_stream = -1;                // wait: the stream the event was recorded on
  int y_first = 0, y_end = 0;           // launch: rows
  const float *launch_dst = nullptr;
};
std::vector<Stream *> streams;
std::map<hipStream_t, Stream *> stream_of;
std::map<hipEvent_t, Event *> events;
std::map<void *, std::pair<int, size_t>> device_mem; // pointer -> (device, bytes)
std::vector<Op> ops;
std::string fail_what; // the next operation of this kind ...
int fail_after = -1;   // ... after this many successful ones fails (-1: none)
int fail_times = 1;    // ... that many times in a row (an allocation is retried once after the geometry cache has been given back)
int wrong_device_calls = 0;

bool should_fail(const char *what) {
  if (fail_what != what || fail_after < 0) return false;
  if (fail_after > 0) {
    --fail_after;
    return false;
  }
  if (--fail_times <= 0) fail_after = -1;
  return true;
}
void reset(int devices, bool reachable) {
  std::lock_guard<std::mutex> l(mu);
  n_devices = devices;
  peers_reachable = reachable;
  ops.clear();
  fail_what.clear();
  fail_after = -1;
  fail_times = 1;
  wrong_device_calls = 0;
  for (Stream *s : streams) s->syncs = 0, s->pending = 0;
}
int owner_of(const void *p) { // device that owns the allocation `p` points into, or -1 (host)
  for (auto &kv : device_mem)
    if ((const char *)p >= (const char *)kv.first && (const char *)p < (const char *)kv.first + kv.second.second) return kv.second.first;
  return -1;
}
} // namespace fake

extern "C" {
hipError_t hipGetDeviceCount(int *n) {
  *n = 8; // (cached by the library at first use: the scenarios use device lists inside 0..7)
  return hipSuccess;
}
hipError_t hipSetDevice(int d) {
  fake::current_device = d;
  return hipSuccess;
}
hipError_t hipGetDevice(int *d) {
  *d = fake::current_device;
  return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipErrorOutOfMemory ? "out of memory (fake)" : "failed (fake)"; }
hipError_t hipMalloc(void **p, size_t n) {
  std::lock_guard<std::mutex> l(fake::mu);
  if (fake::should_fail("malloc")) return hipErrorOutOfMemory;
  *p = std::malloc(n ? n : 1);
  fake::device_mem[*p] = {fake::current_device, n};
  return hipSuccess;
}
hipError_t hipFree(void *p) {
  std::lock_guard<std::mutex> l(fake::mu);
  fake::device_mem.erase(p);
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
  *p = std::calloc(1, n ? n : 1);
  return hipSuccess;
}
hipError_t hipHostFree(void *p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
  *total_b = (size_t)288 << 30;
  *free_b = *total_b;
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
  std::lock_guard<std::mutex> l(fake::mu);
  auto *st = new fake::Stream{fake::current_device, (int)fake::streams.size()};
  fake::streams.push_back(st);
  *s = reinterpret_cast<hipStream_t>(st);
  fake::stream_of[*s] = st;
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *st) {
  *st = hipStreamCaptureStatusNone;
  return hipSuccess;
}
static fake::Stream *stream_checked(hipStream_t s) { // mu held; a stream is used with its device current
  fake::Stream *st = fake::stream_of.count(s) ? fake::stream_of[s] : nullptr;
  if (st && st->device != fake::current_device) ++fake::wrong_device_calls;
  return st;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  std::lock_guard<std::mutex> l(fake::mu);
  if (fake::Stream *st = stream_checked(s)) st->syncs++, st->pending = 0;
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) {
  std::lock_guard<std::mutex> l(fake::mu);
  auto *ev = new fake::Event;
  *e = reinterpret_cast<hipEvent_t>(ev);
  fake::events[*e] = ev;
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  std::lock_guard<std::mutex> l(fake::mu);
  delete fake::events[e];
  fake::events.erase(e);
  return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  std::lock_guard<std::mutex> l(fake::mu);
  fake::Stream *st = stream_checked(s);
  if (st && fake::events.count(e)) {
    fake::events[e]->recorded_on = st->id;
    fake::events[e]->after_op = (int)fake::ops.size() - 1;
    fake::Op op;
    op.what = "record", op.device = fake::current_device, op.stream = st->id;
    fake::ops.push_back(op);
  }
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  std::lock_guard<std::mutex> l(fake::mu);
  fake::Stream *st = stream_checked(s);
  if (st && fake::events.count(e)) {
    fake::Op op;
    op.what = "wait", op.device = fake::current_device, op.stream = st->id, op.event_stream = fake::events[e]->recorded_on;
    fake::ops.push_back(op);
  }
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s) {
  std::lock_guard<std::mutex> l(fake::mu);
  const bool h2d = kind == hipMemcpyHostToDevice;
  if (fake::should_fail(h2d ? "h2d" : "d2h")) return hipErrorUnknown;
  fake::Stream *st = stream_checked(s);
  if (!st) return hipSuccess; // (copies of the geometry cache's list headers on caller streams: not part of this job's record)
  fake::Op op;
  op.what = h2d ? "h2d" : "d2h", op.device = fake::current_device, op.stream = st->id, op.dst = dst, op.src = src, op.bytes = n;
  op.dst_device = fake::owner_of(dst), op.src_device = fake::owner_of(src);
  fake::ops.push_back(op);
  st->pending++;
  return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *dst, int dst_dev, const void *src, int src_dev, size_t n, hipStream_t s) {
  std::lock_guard<std::mutex> l(fake::mu);
  if (fake::should_fail("peer")) return hipErrorUnknown;
  fake::Stream *st = stream_checked(s);
  fake::Op op;
  op.what = "peer", op.device = fake::current_device, op.stream = st ? st->id : -1, op.dst = dst, op.src = src, op.bytes = n;
  op.dst_device = dst_dev, op.src_device = src_dev;
  if (fake::owner_of(dst) != dst_dev || fake::owner_of(src) != src_dev) ++fake::wrong_device_calls;
  fake::ops.push_back(op);
  if (st) st->pending++;
  return hipSuccess;
}
hipError_t hipDeviceCanAccessPeer(int *can, int, int) {
  *can = fake::peers_reachable ? 1 : 0;
  return hipSuccess;
}
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
}

// ---- the kernel launchers and table builders of the .hip units: recorded, not run ---------------------------------------------
namespace lrp {
static hipError_t record_launch(const KParams &P, hipStream_t s) {
  std::lock_guard<std::mutex> l(fake::mu);
  if (fake::should_fail("launch")) return hipErrorUnknown;
  fake::Stream *st = fake::stream_of.count(s) ? fake::stream_of[s] : nullptr;
  if (st && st->device != fake::current_device) ++fake::wrong_device_calls;
  fake::Op op;
  op.what = "launch", op.device = fake::current_device, op.stream = st ? st->id : -1, op.y_first = P.y_offset, op.y_end = P.y_end;
  op.launch_dst = P.dst, op.src = P.src;
  op.src_device = fake::owner_of(P.src), op.dst_device = fake::owner_of(P.dst);
  fake::ops.push_back(op);
  if (st) st->pending++;
  return hipSuccess;
}
hipError_t launch_nearest(const KParams &P, int, int, hipStream_t s) { return record_launch(P, s); }
hipError_t launch_bilinear(const KParams &P, int, int, hipStream_t s) { return record_launch(P, s); }
hipError_t launch_bicubic(const KParams &P, int, int, hipStream_t s) { return record_launch(P, s); }
hipError_t launch_tile_nearest(const KParams &P, int, int, hipStream_t s) { return record_launch(P, s); }
hipError_t launch_tile_bilinear(const KParams &P, int, int, hipStream_t s) { return record_launch(P, s); }
hipError_t launch_tile_bicubic(const KParams &P, int, int, hipStream_t s) { return record_launch(P, s); }
hipError_t launch_win_bicubic(const KParams &P, int, int, hipStream_t s) { return record_launch(P, s); }
hipError_t launch_ss_gather(const KParams &P, int, int, hipStream_t s) { return record_launch(P, s); }
hipError_t launch_corner_fill(const KParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_geo_build_lists(int32_t *, int, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_geo_census(int32_t *, int, int, int, int, bool, hipStream_t) { return hipSuccess; }
hipError_t launch_post_process(float *, uint32_t, int, float, float, hipStream_t) { return hipSuccess; }
hipError_t launch_synth_fill(float *, uint32_t, int, uint32_t, int, hipStream_t) { return hipSuccess; }
hipError_t launch_math_eval(int, const float *, const float *, float *, size_t, hipStream_t) { return hipSuccess; }
hipError_t launch_checksum(const float *, size_t, unsigned long long *, hipStream_t) { return hipSuccess; }
size_t pixel_bytes(int, int channels) { return (size_t)channels * 4; }
hipError_t launch_decode_pixels(const void *, int, int, float *, int, size_t, int, hipStream_t) { return hipSuccess; }
hipError_t launch_encode_pixels(const float *, int, void *, int, int, unsigned, size_t, int, hipStream_t) { return hipSuccess; }
void pixel_tables_host(float decode[256], float threshold[256]) {
  for (int i = 0; i < 256; ++i) decode[i] = threshold[i] = 0.0f;
}
static float g_dummy_table[4];
void TableLease::release() { n = 0; }
hipError_t get_output_tables(int, int, const LensP &, int, int, int, hipStream_t, TableLease &, const float **col_tab, const float **row_tab, bool *plain,
                             int *symmetry) {
  *col_tab = *row_tab = g_dummy_table;
  *plain = false;
  *symmetry = 0;
  return hipSuccess;
}
const float *get_xsep_table(int, const float *, int, int, int, const LensP &, int, int, float, const float *, hipStream_t, TableLease &) { return nullptr; }
void release_output_tables() {}
} // namespace lrp

// ---- scenarios ------------------------------------------------------------------------------------------------------------
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);          \
      return 1;                                                            \
    }                                                                      \
  } while (0)

struct Job {
  lrp_image in;
  std::vector<lrp_image> outs;
  std::vector<float> src;
  std::vector<std::vector<float>> dst;
  std::vector<float> rots;
};
static Job make_job(int n_out, int in_w = 64, int in_h = 32, int face = 40, int channels = 3) {
  Job j;
  std::memset(&j.in, 0, sizeof(j.in));
  j.in.lens.type = LRP_EQUIRECTANGULAR;
  j.in.lens.u.equirectangular = {-1.5707964f, 1.5707964f, -3.1415927f, 3.1415927f};
  j.in.width = in_w, j.in.height = in_h, j.in.channels = channels;
  j.src.assign((size_t)in_w * in_h * channels, 0.5f);
  j.in.data = j.src.data();
  j.dst.resize((size_t)n_out);
  for (int i = 0; i < n_out; ++i) {
    lrp_image o;
    std::memset(&o, 0, sizeof(o));
    o.lens.type = LRP_RECTILINEAR;
    o.lens.u.rectilinear.focal_length = 18.0f;
    o.lens.sensor_width = 36.0f, o.lens.sensor_height = 36.0f;
    o.width = face, o.height = face + i, o.channels = channels; // (outputs of different heights: the bands differ per output)
    j.dst[(size_t)i].assign((size_t)o.width * o.height * channels, -1.0f);
    o.data = j.dst[(size_t)i].data();
    j.outs.push_back(o);
    const float r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    j.rots.insert(j.rots.end(), r, r + 9);
  }
  return j;
}
static int run(Job &j, const std::vector<int> &devices, int interp = LRP_BICUBIC) {
  return lrp_reproject_multi(&j.in, j.outs.data(), (int)j.outs.size(), 1, interp, j.rots.data(), nullptr, devices.data(), (int)devices.size());
}
// every stream that saw an operation of this job has been synchronised after its last one
static bool all_drained() {
  for (fake::Stream *s : fake::streams)
    if (s->pending != 0) return false;
  return true;
}
static int parent_of(int k) {
  int top = 1;
  while (top * 2 <= k) top *= 2;
  return k - top;
}

// n distinct devices (any order of ids): the tree, the events, the devices current, the work split, the drain
static int check_job(const std::vector<int> &devices, int n_out, bool reachable) {
  fake::reset(8, reachable);
  Job j = make_job(n_out);
  CHECK(run(j, devices) == LRP_OK);
  CHECK(fake::wrong_device_calls == 0);
  CHECK(all_drained());
  const int n = (int)devices.size();
  // distinct GPUs in list order = the holders
  std::vector<int> holders;
  for (int d = 0; d < n; ++d)
    if (std::find(devices.begin(), devices.begin() + d, devices[(size_t)d]) == devices.begin() + d) holders.push_back(devices[(size_t)d]);
  // the source: ONE upload (to the first device) and holders - 1 peer copies in tree order — or, unreachable peers, an upload per holder
  std::vector<fake::Op> uploads, peers;
  std::map<int, int> stream_device; // stream id -> device
  for (const fake::Op &op : fake::ops) {
    stream_device[op.stream] = op.device;
    if (op.what == "h2d") uploads.push_back(op);
    if (op.what == "peer") peers.push_back(op);
  }
  const size_t in_bytes = j.src.size() * 4;
  if (reachable) {
    CHECK(uploads.size() == 1 && uploads[0].dst_device == holders[0] && uploads[0].src == j.src.data() && uploads[0].bytes == in_bytes);
    CHECK(peers.size() == holders.size() - 1);
    for (size_t k = 1; k < holders.size(); ++k) {
      const fake::Op &c = peers[k - 1];
      CHECK(c.dst_device == holders[k] && c.src_device == holders[(size_t)parent_of((int)k)] && c.bytes == in_bytes);
      CHECK(c.device == holders[k]); // enqueued with the RECEIVING device current, on its own stream
      // ... behind an event recorded on the parent's stream AFTER the operation that filled the parent's copy
      bool ordered = false;
      for (size_t i = 0; i < fake::ops.size(); ++i) {
        const fake::Op &w = fake::ops[i];
        if (&w == &c || w.what != "wait" || w.stream != c.stream) continue;
        if (stream_device[w.event_stream] == c.src_device) ordered = true;
      }
      CHECK(ordered);
    }
  } else {
    CHECK(peers.empty() && uploads.size() == holders.size());
    for (size_t k = 0; k < holders.size(); ++k) CHECK(uploads[k].dst_device == holders[k] && uploads[k].src == j.src.data());
  }
  // the order within the source's distribution: a parent's copy is enqueued before its child's
  {
    std::map<int, size_t> filled_at; // device -> index of the op that filled it
    for (size_t i = 0; i < fake::ops.size(); ++i) {
      const fake::Op &op = fake::ops[i];
      if (op.what == "h2d" || op.what == "peer") {
        if (op.what == "peer") CHECK(filled_at.count(op.src_device) && filled_at[op.src_device] < i);
        if (!filled_at.count(op.dst_device)) filled_at[op.dst_device] = i;
      }
    }
  }
  // the work: every row of every output rendered once and downloaded once, on the participant the split names
  const bool whole = n_out >= n;
  std::vector<std::vector<int>> rendered((size_t)n_out), downloaded((size_t)n_out);
  for (int i = 0; i < n_out; ++i) rendered[(size_t)i].assign((size_t)j.outs[(size_t)i].height, 0), downloaded[(size_t)i].assign((size_t)j.outs[(size_t)i].height, 0);
  int launches = 0;
  for (size_t i = 0; i < fake::ops.size(); ++i) {
    const fake::Op &op = fake::ops[i];
    if (op.what == "launch") {
      ++launches;
      CHECK(op.src_device == op.device); // reads this GPU's copy of the source (the destination of a band is a virtual image whose origin may lie in front of the buffer)
    }
    if (op.what == "d2h") {
      int which = -1;
      for (int o = 0; o < n_out; ++o)
        if ((const float *)op.dst >= j.dst[(size_t)o].data() && (const float *)op.dst < j.dst[(size_t)o].data() + j.dst[(size_t)o].size()) which = o;
      if (which < 0) { // (the list header of a new geometry-cache entry follows its records to the host: lrp_capi.cpp)
        CHECK(op.bytes == (size_t)lrp::kGeoListHeaderWords * 4 && op.src_device == op.device);
        continue;
      }
      CHECK(op.src_device == op.device);
      const size_t row_floats = (size_t)j.outs[(size_t)which].width * 3;
      const size_t first = (size_t)((const float *)op.dst - j.dst[(size_t)which].data()) / row_floats, rows = op.bytes / 4 / row_floats;
      CHECK(first * row_floats == (size_t)((const float *)op.dst - j.dst[(size_t)which].data()) && rows * row_floats * 4 == op.bytes);
      for (size_t r = first; r < first + rows; ++r) downloaded[(size_t)which][r]++;
      // the launch that rendered these rows is the previous operation of this stream
      const fake::Op *prev = nullptr;
      for (size_t k = i; k-- > 0;)
        if (fake::ops[k].stream == op.stream && fake::ops[k].what == "launch") {
          prev = &fake::ops[k];
          break;
        }
      CHECK(prev != nullptr && prev->y_first == (int)first && prev->y_end == (int)(first + rows));
      for (size_t r = first; r < first + rows; ++r) rendered[(size_t)which][r]++;
      if (whole) CHECK(first == 0 && (int)rows == j.outs[(size_t)which].height && op.device == devices[(size_t)(which % n)]);
    }
  }
  for (int o = 0; o < n_out; ++o)
    for (int r = 0; r < j.outs[(size_t)o].height; ++r) CHECK(rendered[(size_t)o][(size_t)r] == 1 && downloaded[(size_t)o][(size_t)r] == 1);
  CHECK(launches == (whole ? n_out : n_out * n));
  // one synchronisation per participant at the end
  int synced = 0;
  for (fake::Stream *s : fake::streams) synced += s->syncs;
  CHECK(synced == n);
  return 0;
}

static int scenario_tree(int n) {
  std::vector<int> devices;
  for (int d = 0; d < n; ++d) devices.push_back(d);
  if (check_job(devices, 6, true)) return 1;  // whole outputs (n <= 6) or bands (n == 8)
  if (check_job(devices, 1, true)) return 1;  // one output: row bands over all participants
  if (check_job(devices, 11, true)) return 1; // more outputs than participants
  std::vector<int> reversed(devices.rbegin(), devices.rend());
  if (check_job(reversed, 6, true)) return 1; // the root is whichever device the list names first
  if (check_job(devices, 6, false)) return 1; // peers not reachable: an upload per holder, nothing else changes
  return 0;
}

static int scenario_repeats() { // [2 2 5 5 2]: two holders, three participants that read a copy in place
  const std::vector<int> devices = {2, 2, 5, 5, 2};
  if (check_job(devices, 6, true)) return 1;
  if (check_job(devices, 3, true)) return 1;
  // the later occurrences wait for their GPU's copy on their own streams and copy nothing
  int waits_on_holder = 0;
  for (const fake::Op &op : fake::ops)
    if (op.what == "wait") ++waits_on_holder;
  CHECK(waits_on_holder >= 1 + 3); // the peer copy's wait + three later occurrences
  return 0;
}

// a failure in the middle of a job: the status comes back, and every stream has been drained first
static int scenario_failures() {
  const std::vector<int> devices = {0, 1, 2, 3, 4, 5, 6, 7};
  const struct {
    const char *what;
    int after;
  } failures[] = {{"peer", 0}, {"peer", 3}, {"peer", 6}, {"launch", 0}, {"launch", 5}, {"d2h", 0}, {"d2h", 4}, {"h2d", 0}, {"malloc", 0}, {"malloc", 5}};
  for (const auto &f : failures) {
    fake::reset(8, true);
    Job j = make_job(6);
    { // (free the participants' buffers so that `malloc` failures have allocations to fail: a job of another size)
      j = make_job(6, 128 + 8 * (int)(&f - failures), 64);
    }
    fake::fail_what = f.what;
    fake::fail_after = f.after;
    fake::fail_times = std::string(f.what) == "malloc" ? 2 : 1; // (the retry after geo_release_device fails as well)
    const int st = run(j, devices);
    if (st == LRP_OK) std::printf("no failure from %s after %d\n", f.what, f.after);
    CHECK(st != LRP_OK);
    CHECK(st == (std::string(f.what) == "malloc" ? LRP_ERR_OOM : LRP_ERR_HIP));
    CHECK(all_drained());
    CHECK(std::strlen(lrp_last_error()) > 0);
    // ... and the participants are usable again
    fake::reset(8, true);
    Job ok = make_job(6);
    CHECK(run(ok, devices) == LRP_OK && all_drained());
  }
  return 0;
}

// two jobs whose device lists name the same GPUs in opposite orders, at once, many times: one global lock order
static int scenario_lock_order() {
  fake::reset(8, true);
  std::atomic<int> bad{0};
  auto worker = [&](std::vector<int> devices) {
    for (int i = 0; i < 60; ++i) {
      Job j = make_job(6);
      if (run(j, devices) != LRP_OK) ++bad;
    }
  };
  std::thread a(worker, std::vector<int>{0, 1, 2, 3}), b(worker, std::vector<int>{3, 2, 1, 0}), c(worker, std::vector<int>{2, 0, 2, 1}), d(worker, std::vector<int>{5, 6});
  a.join(), b.join(), c.join(), d.join();
  CHECK(bad == 0 && fake::wrong_device_calls == 0 && all_drained());
  return 0;
}

static int scenario_arguments() {
  fake::reset(8, true);
  Job j = make_job(2);
  const int none[1] = {0};
  CHECK(lrp_reproject_multi(&j.in, j.outs.data(), 2, 1, LRP_BICUBIC, j.rots.data(), nullptr, nullptr, 1) == LRP_ERR_BAD_ARG);
  CHECK(lrp_reproject_multi(&j.in, j.outs.data(), 2, 1, LRP_BICUBIC, j.rots.data(), nullptr, none, 0) == LRP_ERR_BAD_ARG);
  const int beyond[2] = {0, 8};
  CHECK(lrp_reproject_multi(&j.in, j.outs.data(), 2, 1, LRP_BICUBIC, j.rots.data(), nullptr, beyond, 2) == LRP_ERR_NO_DEVICE);
  CHECK(fake::ops.empty()); // nothing was enqueued
  const int one[1] = {3};
  CHECK(lrp_reproject_multi(&j.in, j.outs.data(), 0, 1, LRP_BICUBIC, j.rots.data(), nullptr, one, 1) == LRP_OK && fake::ops.empty());
  CHECK(lrp_reproject_multi(&j.in, j.outs.data(), 2, 0, LRP_BICUBIC, j.rots.data(), nullptr, one, 1) == LRP_OK && fake::ops.empty()); // num_samples 0: the reference's loop body never runs
  j.outs[1].lens.type = LRP_FISHEYE_EQUISOLID;
  CHECK(lrp_reproject_multi(&j.in, j.outs.data(), 2, 1, LRP_BICUBIC, j.rots.data(), nullptr, one, 1) == LRP_ERR_OUTPUT_LENS && fake::ops.empty());
  return 0;
}

int main(int argc, char **argv) {
  const std::string name = argc > 1 ? argv[1] : "";
  int rc = 2;
  if (name == "tree2") rc = scenario_tree(2);
  if (name == "tree3") rc = scenario_tree(3);
  if (name == "tree8") rc = scenario_tree(8);
  if (name == "repeats") rc = scenario_repeats();
  if (name == "failures") rc = scenario_failures();
  if (name == "lock_order") rc = scenario_lock_order();
  if (name == "arguments") rc = scenario_arguments();
  if (rc == 0) std::printf("ok %s\n", name.c_str());
  if (rc == 2) std::printf("unknown scenario\n");
  return rc;
}
