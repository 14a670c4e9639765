// lrp_tile_win.hip — bicubic window-kernel instantiations (lrp_kernel_v2.h): RGBA, plain blocks.
#include "lrp_kernel_v2.h"

namespace lrp {
hipError_t launch_win_bicubic_quad(const KParams &P, int out_idx, int in_mode, hipStream_t stream);     // lrp_tile_winq.hip
hipError_t launch_win_bicubic_rgb(const KParams &P, int out_idx, int in_mode, hipStream_t stream);      // lrp_tile_win3.hip
hipError_t launch_win_bicubic_rgb_quad(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winq3.hip
hipError_t launch_win_bicubic_rgbaz(const KParams &P, int out_idx, int in_mode, hipStream_t stream);      // lrp_tile_win5.hip
hipError_t launch_win_bicubic_rgbaz_quad(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winq5.hip
// P.channels must be 3, 4 or 5, P.num_samples 1.
hipError_t launch_win_bicubic(const KParams &P, int out_idx, int in_mode, hipStream_t stream) {
  if (P.channels == 5)
    return P.quad == 1 ? launch_win_bicubic_rgbaz_quad(P, out_idx, in_mode, stream) : launch_win_bicubic_rgbaz(P, out_idx, in_mode, stream);
  if (P.channels == 3)
    return P.quad == 1 ? launch_win_bicubic_rgb_quad(P, out_idx, in_mode, stream) : launch_win_bicubic_rgb(P, out_idx, in_mode, stream);
  if (P.quad == 1) return launch_win_bicubic_quad(P, out_idx, in_mode, stream);
  return launch_win_bicubic_impl<false, 4>(P, out_idx, in_mode, stream);
}
} // namespace lrp
