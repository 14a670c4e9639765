// lrp_tile_win.hip — bicubic window-kernel instantiations (lrp_kernel_v2.h).
#include "lrp_kernel_v2.h"

namespace lrp {
hipError_t launch_win_bicubic(const KParams &P, int out_idx, int in_mode, hipStream_t stream) {
  return launch_win_bicubic_impl(P, out_idx, in_mode, stream);
}
#if LRP_ABLATE == 5
extern "C" int lrp_debug_read_stamps(unsigned long long *out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lrp_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lrp_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
} // namespace lrp
