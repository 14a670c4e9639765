// lrp_tables.h — cached separable output-lens terms (see lrp_tables.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "lrp_params.h"

namespace lrp {
// Keeps the tables a lookup returned alive until the kernel reading them has been enqueued:
// the lookups pin their cache entries, release() (or the destructor) unpins them.  The cache
// only ever frees unpinned entries, and only after synchronising their device, which waits for
// every launch enqueued before the unpin (see lrp_tables.hip).
struct TableLease {
  std::atomic<int> *pins[2] = {nullptr, nullptr};
  int n = 0;
  TableLease() = default;
  TableLease(const TableLease &) = delete;
  TableLease &operator=(const TableLease &) = delete;
  ~TableLease() { release(); }
  void release();
};
// out_lens: kRect or kEquirect.  On success *col_tab has 2 * out_w * ns floats
// (rectilinear uses the first half only) and *row_tab has out_h * ns floats, both
// complete (the build is synchronous) and valid while `lease` pins them.
// *plain: no table value is -0.0f, an infinity or a NaN (then an identity rotation
// matrix changes no bit of any ray and may be dropped).  *symmetry (ns == 1 only, else 0): bit 0 = the columns
// are mirror images bit for bit, vx(W-1-x) == -vx(x), vz(W-1-x) == vz(x); bit 1 = the rows are,
// vy(H-1-y) == -vy(y); and no |vx| / |vy| other than a centre zero is below 2^-60.
// A miss builds the tables on `stream` and waits for it.
hipError_t get_output_tables(int device, int out_lens, const LensP &lens, int out_w, int out_h, int ns, hipStream_t stream,
                             TableLease &lease, const float **col_tab, const float **row_tab, bool *plain, int *symmetry);
// Column-separable source x for a rectilinear / equirectangular source (in_mode kInRect,
// kInEquirect or kInEquirectLoop) behind the output tables `col_tab`: [3][out_w * ns] floats
// (rotated ray x, rotated ray z, source texel x), or null when it does not apply (the sign
// of the row term matters in some column, cache full, no memory).  rot: row-major 3x3 whose
// [0][1] and [2][1] entries the caller has checked to be zeros, or null.
const float *get_xsep_table(int device, const float *col_tab, int out_lens, int out_w, int ns, const LensP &in_lens,
                            int in_mode, int in_w, float in_lon_span, const float *rot, hipStream_t stream,
                            TableLease &lease);
void release_output_tables();
} // namespace lrp
