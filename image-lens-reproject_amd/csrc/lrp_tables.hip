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
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "lrp_device.h"
#include "lrp_tables.h"

namespace lrp {

namespace {

struct TableArgs {
  LensP lens;
  int32_t out_lens; // kRect or kEquirect
  int32_t out_w, out_h, ns;
  float *tab;
};

__global__ __launch_bounds__(256) void build_tables_kernel(const TableArgs A) {
  const int n_col = A.out_w * A.ns, n_row = A.out_h * A.ns;
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= n_col + n_row) return;
  const bool is_col = i < n_col;
  const int j = is_col ? i : i - n_col;
  const int pix = j / A.ns, ss = j - pix * A.ns;
  const float extent = (float)(is_col ? A.out_w : A.out_h);
  // src/reproject.cpp:287-288,295,298
  const float c = ((float)pix + 0.5f) - extent * 0.5f;
  const float sc = c + ((float)ss + 1.0f) / ((float)A.ns + 1.0f) - 0.5f;
  const LensP &L = A.lens;
  if (A.out_lens == kRect) {
    const float focal = L.p[0];
    if (is_col)
      A.tab[j] = sc / extent * L.sensor_width / focal; // :155
    else
      A.tab[2 * n_col + j] = sc / extent * L.sensor_height / focal; // :156
  } else {
    const float lat_min = L.p[0], lat_max = L.p[1], lon_min = L.p[2], lon_max = L.p[3];
    if (is_col) {
      const float lon_span = lon_max - lon_min;
      const float lon = ((sc / extent) + 0.5f) * lon_span + lon_min; // :251
      float sn, cs;
      sincosf_(lon, sn, cs);
      A.tab[j] = sn;          // :254
      A.tab[n_col + j] = -cs; // :255
    } else {
      const float lat_span = lat_max - lat_min;
      const float lat = ((sc / extent) + 0.5f) * lat_span + lat_min; // :252
      A.tab[2 * n_col + j] = sinf_(lat);                           // :256
    }
  }
}

struct Entry {
  int device;
  TableArgs key; // tab = device pointer of the finished tables
};

std::mutex g_mutex;
std::vector<Entry> g_entries;
constexpr size_t kMaxEntries = 256;

bool same_key(const TableArgs &a, const TableArgs &b) {
  return a.out_lens == b.out_lens && a.out_w == b.out_w && a.out_h == b.out_h && a.ns == b.ns &&
         std::memcmp(&a.lens, &b.lens, sizeof(LensP)) == 0;
}

} // namespace

hipError_t get_output_tables(int device, int out_lens, const LensP &lens, int out_w, int out_h, int ns,
                             const float **col_tab, const float **row_tab) {
  *col_tab = *row_tab = nullptr;
  TableArgs want;
  std::memset(&want, 0, sizeof(want));
  want.lens = lens;
  want.out_lens = out_lens;
  want.out_w = out_w;
  want.out_h = out_h;
  want.ns = ns;
  const size_t n_col = (size_t)out_w * ns, n_row = (size_t)out_h * ns;
  std::lock_guard<std::mutex> lock(g_mutex);
  for (const Entry &e : g_entries)
    if (e.device == device && same_key(e.key, want)) {
      *col_tab = e.key.tab;
      *row_tab = e.key.tab + 2 * n_col;
      return hipSuccess;
    }
  if (g_entries.size() >= kMaxEntries) return hipErrorOutOfMemory; // caller falls back to the per-pixel kernel
  // Miss: build synchronously on the legacy default stream (host blocks once per
  // new output-lens configuration; do this before capturing a hipGraph).
  float *tab = nullptr;
  hipError_t e = hipMalloc(&tab, (2 * n_col + n_row) * sizeof(float));
  if (e != hipSuccess) return e;
  want.tab = tab;
  const unsigned blocks = (unsigned)((n_col + n_row + 255) / 256);
  hipLaunchKernelGGL(build_tables_kernel, dim3(blocks), dim3(256), 0, 0, want);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(0);
  if (e != hipSuccess) {
    (void)hipFree(tab);
    return e;
  }
  g_entries.push_back(Entry{device, want});
  *col_tab = tab;
  *row_tab = tab + 2 * n_col;
  return hipSuccess;
}

void release_output_tables() {
  std::lock_guard<std::mutex> lock(g_mutex);
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (const Entry &e : g_entries) {
    (void)hipSetDevice(e.device);
    (void)hipDeviceSynchronize();
    (void)hipFree(e.key.tab);
  }
  g_entries.clear();
  (void)hipSetDevice(cur);
}

} // namespace lrp
