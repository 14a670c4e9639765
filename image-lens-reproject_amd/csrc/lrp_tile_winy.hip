// lrp_tile_winy.hip — bicubic window-kernel instantiations (lrp_kernel_v2.h): RGBA, mirrored rows (pan rotations).
#include "lrp_kernel_v2.h"

namespace lrp {
hipError_t launch_win_bicubic_c4_m2(const KParams &P, int out_idx, int in_mode, hipStream_t stream) {
  return launch_win_bicubic_impl<2, 4>(P, out_idx, in_mode, stream);
}
} // namespace lrp
