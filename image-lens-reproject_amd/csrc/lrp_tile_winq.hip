// lrp_tile_winq.hip — bicubic window-kernel instantiations (lrp_kernel_v2.h): RGBA, mirrored in both axes.
#include "lrp_kernel_v2.h"

namespace lrp {
hipError_t launch_win_bicubic_c4_m1(const KParams &P, int out_idx, int in_mode, hipStream_t stream) {
  return launch_win_bicubic_impl<1, 4>(P, out_idx, in_mode, stream);
}
} // namespace lrp

#if defined(LRP_TIER_STATS)
extern "C" void lrp_debug_read_tiers(unsigned out[8]) { // mirrored RGBA instantiations only
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lrp::g_tier_stats), sizeof(unsigned) * 8);
  unsigned zero[8] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(lrp::g_tier_stats), zero, sizeof(zero));
}
#endif
