// lrp_tables.h — cached separable output-lens terms (see lrp_tables.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "lrp_params.h"

namespace lrp {
// out_lens: kRect or kEquirect.  On success *col_tab has 2 * out_w * ns floats
// (rectilinear uses the first half only) and *row_tab has out_h * ns floats, both
// complete (the build is synchronous) and valid until release_output_tables().
// *plain: no table value is -0.0f, an infinity or a NaN (then an identity rotation
// matrix changes no bit of any ray and may be dropped).  *mirror: ns == 1 and the tables are
// symmetric about the image centre bit for bit: vx(W-1-x) == -vx(x), vz(W-1-x) == vz(x),
// vy(H-1-y) == -vy(y).
hipError_t get_output_tables(int device, int out_lens, const LensP &lens, int out_w, int out_h, int ns,
                             const float **col_tab, const float **row_tab, bool *plain, bool *mirror);
// Column-separable source x for a rectilinear / equirectangular source (in_mode kInRect,
// kInEquirect or kInEquirectLoop) behind the output tables `col_tab`: [3][out_w * ns] floats
// (rotated ray x, rotated ray z, source texel x), or null when it does not apply (the sign
// of the row term matters in some column, cache full, no memory).  rot: row-major 3x3 whose
// [0][1] and [2][1] entries the caller has checked to be zeros, or null.
const float *get_xsep_table(int device, const float *col_tab, int out_lens, int out_w, int ns, const LensP &in_lens,
                            int in_mode, int in_w, float in_lon_span, const float *rot);
void release_output_tables();
} // namespace lrp
