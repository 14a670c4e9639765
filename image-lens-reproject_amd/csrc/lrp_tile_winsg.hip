// lrp_tile_winsg.hip — bicubic window-kernel instantiations (lrp_kernel_v2.h): RGBA, num_samples 2-4 from a geometry-cache entry of sub-samples.
#include "lrp_kernel_v2.h"

namespace lrp {
hipError_t launch_win_bicubic_ssg_c4(const KParams &P, int out_idx, int in_mode, hipStream_t stream) {
  return launch_win_bicubic_impl<0, 4, true, true>(P, out_idx, in_mode, stream);
}
} // namespace lrp
