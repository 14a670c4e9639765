// lrp_tables.hip — separable output-lens terms for the tile kernel.
//
// rectilinear_to_vec (src/reproject.cpp:152-158) and equirectangular_to_vec
// (:245-257) are separable: x (and z) depend on the output column and the
// horizontal sub-sample only, y on the row and the vertical sub-sample only.
// The reference recomputes them for every pixel (4 IEEE divides, or a
// double-precision sincosf + sinf).  Here they are evaluated once per column and
// once per row by build_tables_kernel — the very same operations in the same
// order on the same inputs, so the same bits — and the tile kernel just loads
// them.  Tables are cached per (device, output lens, output size, num_samples):
// a batch of images with one output lens builds them once.
//
// Column-separable source x (build_xsep_kernel).  The rotated ray is n = R v with
// v = (vx(column), vy(row), vz(column)).  When R[0][1] and R[2][1] are zeros — no
// rotation, the identity, any pan-only rotation (the cubemap's side faces) — nx
// and nz are functions of the column alone: the term R[0][1] * vy is a zero whose
// sign is the sign of vy, and adding it changes nothing unless the rest of the sum
// is itself a zero.  The builder evaluates nx, nz for vy = +1 and vy = -1 with the
// per-pixel expression; only if both agree bit for bit in every column is the
// table used.  For a rectilinear or equirectangular SOURCE the horizontal source
// coordinate depends on nx and nz only (src/reproject.cpp:163,165 / :262,268), so
// it is evaluated once per column here — same functions, same operations — and a
// pixel is left with the vertical half of the projection.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "lrp_device.h"
#include "lrp_source_axes.h"
#include "lrp_tables.h"

namespace lrp {

namespace {

struct TableArgs {
  LensP lens;
  int32_t out_lens; // kRect or kEquirect
  int32_t out_w, out_h, ns;
  float *tab;
  int *flags; // see build_tables_kernel
};


// One table value: is_col ? (column j: vx, and vz for the equirectangular target) : (row j: vy).
__device__ __forceinline__ void table_value(const TableArgs &A, bool is_col, int j, float &v0, float &v1) {
  const int pix = j / A.ns, ss = j - pix * A.ns;
  const float extent = (float)(is_col ? A.out_w : A.out_h);
  // src/reproject.cpp:287-288,295,298
  const float c = ((float)pix + 0.5f) - extent * 0.5f;
  const float sc = c + ((float)ss + 1.0f) / ((float)A.ns + 1.0f) - 0.5f;
  const LensP &L = A.lens;
  v1 = 0.0f;
  if (A.out_lens == kRect) {
    const float focal = L.p[0];
    v0 = is_col ? sc / extent * L.sensor_width / focal    // :155
                : sc / extent * L.sensor_height / focal; // :156
  } else {
    const float lat_min = L.p[0], lat_max = L.p[1], lon_min = L.p[2], lon_max = L.p[3];
    if (is_col) {
      const float lon_span = lon_max - lon_min;
      const float lon = ((sc / extent) + 0.5f) * lon_span + lon_min; // :251
      float sn, cs;
      sincosf_(lon, sn, cs);
      v0 = sn;  // :254
      v1 = -cs; // :255
    } else {
      const float lat_span = lat_max - lat_min;
      const float lat = ((sc / extent) + 0.5f) * lat_span + lat_min; // :252
      v0 = sinf_(lat);                                             // :256
    }
  }
}

// flags[0]: bit 0 = some table value is -0.0f, an infinity or a NaN; bit 1 = the columns are
// not mirror images of each other (vx(W-1-j) == -vx(j), vz(W-1-j) == vz(j), bit for bit);
// bit 2 = the rows are not (vy(H-1-j) == -vy(j)).  Mirror bits are only meaningful for ns == 1.  A vx / vy of a
// magnitude below 2^-60 (other than the exact zero of a centre column / row) also clears the axis' symmetry: the
// kernels that share work between mirror images multiply these by matrix entries and must not meet an underflow.
__global__ __launch_bounds__(256) void build_tables_kernel(const TableArgs A) {
  const int n_col = A.out_w * A.ns, n_row = A.out_h * A.ns;
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= n_col + n_row) return;
  const bool is_col = i < n_col;
  const int j = is_col ? i : i - n_col;
  float v0, v1, m0, m1;
  table_value(A, is_col, j, v0, v1);
  table_value(A, is_col, (is_col ? n_col : n_row) - 1 - j, m0, m1);
  int flags = 0;
  auto note = [&](float v) {
    const uint32_t b = __float_as_uint(v);
    if (b == 0x80000000u || (b & 0x7f800000u) == 0x7f800000u) flags |= 1;
  };
  note(v0);
  // the centre column / row of an odd-sized image is its own mirror image (its +0 has no -0 partner)
  const bool self = 2 * j == (is_col ? n_col : n_row) - 1;
  if (!self && __float_as_uint(m0) != (__float_as_uint(v0) ^ 0x80000000u)) flags |= is_col ? 2 : 4;
  if (!self && __builtin_fabsf(v0) < 0x1p-60f) flags |= is_col ? 2 : 4;
  // ... and the centre column / row, its own mirror image, must be the zero it is for a symmetric lens (a one-pixel
  // wide or high target of an asymmetric partial panorama has nothing but a centre)
  if (self && v0 != 0.0f) flags |= is_col ? 2 : 4;
  if (is_col) {
    A.tab[j] = v0;
    if (A.out_lens != kRect) {
      A.tab[n_col + j] = v1;
      note(v1);
      if (__float_as_uint(m1) != __float_as_uint(v1)) flags |= 2;
    }
  } else {
    A.tab[2 * n_col + j] = v0;
  }
  if (flags) atomicOr(A.flags, flags);
}

struct XsepArgs {
  const float *col_tab; // of the output lens
  int32_t out_lens, n;  // n = out_w * ns
  LensP in_lens;
  int32_t in_mode, in_w;
  float in_lon_span;
  int32_t has_rot;
  float rx[3], rz[3]; // rows 0 and 2 of the rotation matrix
  float *tab;         // [3][n]: nx, nz, source texel x
  int *flags;         // flags[0] != 0: the two signs of vy disagree somewhere
};

__global__ __launch_bounds__(256) void build_xsep_kernel(const XsepArgs A) {
  const int j = (int)(blockIdx.x * 256 + threadIdx.x);
  if (j >= A.n) return;
  const float vx = A.col_tab[j];
  const float vz = A.out_lens == kRect ? -1.0f : A.col_tab[A.n + j];
  const LensP &L = A.in_lens;
  const float img_w = (float)A.in_w;
  float nx[2], nz[2], sx[2];
  for (int cls = 0; cls < 2; ++cls) {
    const float vy = cls ? -1.0f : 1.0f;
    nx[cls] = vx;
    nz[cls] = vz;
    if (A.has_rot) { // src/reproject.cpp:303-311, rows 0 and 2
      nx[cls] = A.rx[0] * vx + A.rx[1] * vy + A.rx[2] * vz;
      nz[cls] = A.rz[0] * vx + A.rz[1] * vy + A.rz[2] * vz;
    }
    float px;
    if (A.in_mode == kInRect)
      px = rect_axis(nx[cls] / -nz[cls], img_w, L.sensor_width, L.p[0]); // :163, :165
    else
      px = equirect_cx(nx[cls], nz[cls], L.p[2], A.in_lon_span, img_w);  // :262, :268
    sx[cls] = texel_coord(px, img_w);                                     // :323
  }
  if (__float_as_uint(nx[0]) != __float_as_uint(nx[1]) || __float_as_uint(nz[0]) != __float_as_uint(nz[1]) ||
      __float_as_uint(sx[0]) != __float_as_uint(sx[1]))
    atomicOr(A.flags, 1);
  A.tab[j] = nx[0];
  A.tab[A.n + j] = nz[0];
  A.tab[2 * A.n + j] = sx[0];
}

struct Entry {
  int device;
  TableArgs key; // tab = device pointer of the finished tables
  int flags;     // as left by build_tables_kernel
  std::atomic<int> pins{0}; // leases handed out and not yet released (see TableLease)
};

struct XsepEntry {
  int device;
  XsepArgs key;  // tab = device pointer of the finished table (null: the rotation does not separate)
  std::atomic<int> pins{0};
};
// unique_ptr: a lease holds the address of an entry's pin counter across vector growth
std::vector<std::unique_ptr<XsepEntry>> g_xsep;

bool same_xsep(const XsepArgs &a, const XsepArgs &b) {
  return a.col_tab == b.col_tab && a.out_lens == b.out_lens && a.n == b.n && a.in_mode == b.in_mode && a.in_w == b.in_w &&
         a.has_rot == b.has_rot && std::memcmp(&a.in_lens, &b.in_lens, sizeof(LensP)) == 0 &&
         std::memcmp(&a.in_lon_span, &b.in_lon_span, sizeof(float)) == 0 && std::memcmp(a.rx, b.rx, sizeof(a.rx)) == 0 &&
         std::memcmp(a.rz, b.rz, sizeof(a.rz)) == 0;
}

std::mutex g_mutex;
std::vector<std::unique_ptr<Entry>> g_entries;
constexpr size_t kMaxEntries = 256;

// Table lifetime.  A lookup pins the entry it returns (under g_mutex); the caller releases the
// pin once the kernel that reads the table has been ENQUEUED (TableLease, lrp_tables.h).
// Eviction — under g_mutex, so no new pin can appear meanwhile — only considers entries
// whose pin count is zero: every launch that ever used such an entry is already in some
// stream's queue, and the device synchronisation that precedes the hipFree waits for it.
// A pinned entry (a launch is about to be enqueued by another thread) is never freed.
// A pinned column table implies a pinned output table: both pins belong to one lease.
// (hipGraphs captured earlier keep pointing at freed tables — same contract as
// lrp_release_cached_tables.)
void sync_device(int device) {
  (void)hipSetDevice(device);
  (void)hipDeviceSynchronize();
}

void drop_xsep_where(const std::vector<const float *> *parents, size_t max_drop) { // g_mutex held
  size_t dropped = 0;
  for (size_t i = 0; i < g_xsep.size() && dropped < max_drop;) {
    XsepEntry &x = *g_xsep[i];
    bool match = parents == nullptr;
    if (parents)
      for (const float *t : *parents) match = match || x.key.col_tab == t;
    if (match && x.pins.load(std::memory_order_acquire) == 0) {
      if (x.key.tab) {
        sync_device(x.device);
        (void)hipFree(x.key.tab);
      }
      g_xsep.erase(g_xsep.begin() + (long)i);
      ++dropped;
    } else {
      ++i;
    }
  }
}

// Cache full: drop the older half of the (unpinned) output tables together with the column
// tables built on them.  A table build blocks the calling thread anyway; a process that keeps
// producing new output geometries pays one device synchronisation per 128 of them.
void evict_older_half() { // g_mutex held
  int cur = 0;
  (void)hipGetDevice(&cur);
  const size_t n_drop = g_entries.size() / 2;
  std::vector<const float *> dropped;
  for (size_t i = 0; i < g_entries.size() && dropped.size() < n_drop; ++i)
    if (g_entries[i]->pins.load(std::memory_order_acquire) == 0) dropped.push_back(g_entries[i]->key.tab);
  // column tables first (they were built from, and are keyed on, the output tables)
  drop_xsep_where(&dropped, g_xsep.size());
  for (size_t i = 0; i < g_entries.size();) {
    Entry &en = *g_entries[i];
    bool gone = false;
    for (const float *t : dropped) gone = gone || en.key.tab == t;
    if (gone) {
      sync_device(en.device);
      (void)hipFree(en.key.tab);
      g_entries.erase(g_entries.begin() + (long)i);
    } else {
      ++i;
    }
  }
  (void)hipSetDevice(cur);
}

void evict_older_xsep_half() { // g_mutex held
  int cur = 0;
  (void)hipGetDevice(&cur);
  drop_xsep_where(nullptr, g_xsep.size() / 2);
  (void)hipSetDevice(cur);
}

bool same_key(const TableArgs &a, const TableArgs &b) { // tab / flags are results, not part of the key
  return a.out_lens == b.out_lens && a.out_w == b.out_w && a.out_h == b.out_h && a.ns == b.ns &&
         std::memcmp(&a.lens, &b.lens, sizeof(LensP)) == 0;
}

// The build runs on the caller's stream (no implicit synchronisation with other streams of the
// device through the legacy default stream); the host waits for the flag word, i.e. once per new
// output-lens configuration.  Do this before capturing a hipGraph.
hipError_t read_flag(const int *d_flag, int *flag, hipStream_t stream) {
  hipError_t e = hipMemcpyAsync(flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  return e;
}

} // namespace

void TableLease::release() {
  if (n == 0) return;
  // Under g_mutex and child first (pins[1], the xsep table built on the output table pins[0]): an eviction pass never
  // sees a pinned column table whose output table has already been unpinned — it would free the parent and keep an
  // xsep entry keyed on the dangling pointer, which a later hipMalloc may hand out again for another lens.
  std::lock_guard<std::mutex> lock(g_mutex);
  for (int i = n - 1; i >= 0; --i) pins[i]->fetch_sub(1, std::memory_order_release);
  n = 0;
}

hipError_t get_output_tables(int device, int out_lens, const LensP &lens, int out_w, int out_h, int ns, hipStream_t stream,
                             TableLease &lease, const float **col_tab, const float **row_tab, bool *plain, int *symmetry) {
  *col_tab = *row_tab = nullptr;
  *plain = false;
  *symmetry = 0;
  TableArgs want;
  std::memset(&want, 0, sizeof(want));
  want.lens = lens;
  want.out_lens = out_lens;
  want.out_w = out_w;
  want.out_h = out_h;
  want.ns = ns;
  const size_t n_col = (size_t)out_w * ns, n_row = (size_t)out_h * ns;
  std::lock_guard<std::mutex> lock(g_mutex);
  auto hand_out = [&](Entry &e) {
    e.pins.fetch_add(1, std::memory_order_acq_rel);
    lease.pins[lease.n++] = &e.pins;
    *col_tab = e.key.tab;
    *row_tab = e.key.tab + 2 * n_col;
    *plain = !(e.flags & 1);
    *symmetry = ns == 1 ? ((e.flags & 2) ? 0 : 1) | ((e.flags & 4) ? 0 : 2) : 0;
  };
  for (const auto &e : g_entries)
    if (e->device == device && same_key(e->key, want)) {
      hand_out(*e);
      return hipSuccess;
    }
  if (g_entries.size() >= kMaxEntries) evict_older_half();
  float *tab = nullptr;
  hipError_t e = hipMalloc(&tab, (2 * n_col + n_row + 1) * sizeof(float)); // + one flag word
  if (e != hipSuccess) return e;
  want.tab = tab;
  want.flags = reinterpret_cast<int *>(tab + 2 * n_col + n_row);
  int flag = 7;
  e = hipMemsetAsync(want.flags, 0, sizeof(int), stream);
  const unsigned blocks = (unsigned)((n_col + n_row + 255) / 256);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(build_tables_kernel, dim3(blocks), dim3(256), 0, stream, want);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = read_flag(want.flags, &flag, stream); // also waits for the build
  if (e != hipSuccess) {
    (void)hipFree(tab);
    return e;
  }
  g_entries.emplace_back(new Entry{device, want, flag});
  hand_out(*g_entries.back());
  return hipSuccess;
}

const float *get_xsep_table(int device, const float *col_tab, int out_lens, int out_w, int ns, const LensP &in_lens,
                            int in_mode, int in_w, float in_lon_span, const float *rot, hipStream_t stream,
                            TableLease &lease) {
  XsepArgs want;
  std::memset(&want, 0, sizeof(want));
  want.col_tab = col_tab;
  want.out_lens = out_lens;
  want.n = out_w * ns;
  want.in_lens = in_lens;
  want.in_mode = in_mode == kInRect ? kInRect : kInEquirect; // wrapping does not enter the coordinate
  want.in_w = in_w;
  want.in_lon_span = in_lon_span;
  want.has_rot = rot != nullptr;
  if (rot) {
    std::memcpy(want.rx, rot, sizeof(want.rx));
    std::memcpy(want.rz, rot + 6, sizeof(want.rz));
  }
  std::lock_guard<std::mutex> lock(g_mutex);
  auto hand_out = [&](XsepEntry &e) -> const float * {
    if (!e.key.tab) return nullptr; // "does not separate": nothing to keep alive
    e.pins.fetch_add(1, std::memory_order_acq_rel);
    lease.pins[lease.n++] = &e.pins;
    return e.key.tab;
  };
  for (const auto &e : g_xsep)
    if (e->device == device && same_xsep(e->key, want)) return hand_out(*e);
  if (g_xsep.size() >= kMaxEntries) evict_older_xsep_half();
  float *tab = nullptr;
  if (hipMalloc(&tab, (3 * (size_t)want.n + 1) * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  want.tab = tab;
  want.flags = reinterpret_cast<int *>(tab + 3 * (size_t)want.n);
  int flag = 1;
  hipError_t e = hipMemsetAsync(want.flags, 0, sizeof(int), stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(build_xsep_kernel, dim3((unsigned)((want.n + 255) / 256)), dim3(256), 0, stream, want);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = read_flag(want.flags, &flag, stream);
  if (e != hipSuccess || flag != 0) { // failed, or the sign of vy matters in some column: remember "does not separate"
    (void)hipGetLastError();
    (void)hipFree(tab);
    want.tab = nullptr;
    want.flags = nullptr;
    if (e != hipSuccess) return nullptr;
  }
  g_xsep.emplace_back(new XsepEntry{device, want});
  return hand_out(*g_xsep.back());
}

// Frees every table no lease pins (callers: tests, shutdown paths; not concurrently with launches
// that are being enqueued — those keep their tables, which stay cached).
void release_output_tables() {
  std::lock_guard<std::mutex> lock(g_mutex);
  int cur = 0;
  (void)hipGetDevice(&cur);
  drop_xsep_where(nullptr, g_xsep.size());
  for (size_t i = 0; i < g_entries.size();) {
    Entry &en = *g_entries[i];
    bool child = false;
    for (const auto &x : g_xsep) child = child || x->key.col_tab == en.key.tab;
    if (en.pins.load(std::memory_order_acquire) == 0 && !child) {
      sync_device(en.device);
      (void)hipFree(en.key.tab);
      g_entries.erase(g_entries.begin() + (long)i);
    } else {
      ++i;
    }
  }
  (void)hipSetDevice(cur);
}

} // namespace lrp
