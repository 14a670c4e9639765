// lrp_tables.h — cached separable output-lens terms (see lrp_tables.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "lrp_params.h"

namespace lrp {
// out_lens: kRect or kEquirect.  On success *col_tab has 2 * out_w * ns floats
// (rectilinear uses the first half only) and *row_tab has out_h * ns floats, both
// complete (the build is synchronous) and valid until release_output_tables().
hipError_t get_output_tables(int device, int out_lens, const LensP &lens, int out_w, int out_h, int ns,
                             const float **col_tab, const float **row_tab);
void release_output_tables();
} // namespace lrp
