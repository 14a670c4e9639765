// lrp_tile_winr.hip — bicubic window-kernel instantiations (lrp_kernel_v2.h): RGBA, shared rays (equidistant target, any rotation).
#include "lrp_kernel_v2.h"

namespace lrp {
hipError_t launch_win_bicubic_c4_m4(const KParams &P, int out_idx, int in_mode, hipStream_t stream) {
  return launch_win_bicubic_impl<4, 4>(P, out_idx, in_mode, stream);
}
} // namespace lrp
