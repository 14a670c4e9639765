// lrp_tile_win.hip — bicubic window-kernel instantiations (lrp_kernel_v2.h), plain blocks.
#include "lrp_kernel_v2.h"

namespace lrp {
hipError_t launch_win_bicubic_quad(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winq.hip
hipError_t launch_win_bicubic(const KParams &P, int out_idx, int in_mode, hipStream_t stream) {
  if (P.quad == 1) return launch_win_bicubic_quad(P, out_idx, in_mode, stream);
  return launch_win_bicubic_impl<false>(P, out_idx, in_mode, stream);
}
} // namespace lrp
