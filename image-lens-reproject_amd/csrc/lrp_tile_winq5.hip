// lrp_tile_winq5.hip — bicubic window-kernel instantiations (lrp_kernel_v2.h): RGBAZ, mirrored blocks.
#include "lrp_kernel_v2.h"

namespace lrp {
hipError_t launch_win_bicubic_rgbaz_quad(const KParams &P, int out_idx, int in_mode, hipStream_t stream) {
  return launch_win_bicubic_impl<true, 5>(P, out_idx, in_mode, stream);
}
} // namespace lrp
