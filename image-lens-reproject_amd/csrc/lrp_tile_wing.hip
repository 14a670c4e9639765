// lrp_tile_wing.hip — bicubic window-kernel instantiations (lrp_kernel_v2.h): RGBA, coordinates from the geometry cache (GeoRead).
#include "lrp_kernel_v2.h"

namespace lrp {
hipError_t launch_win_bicubic_geo_c4(const KParams &P, int out_idx, int in_mode, hipStream_t stream) {
  return launch_win_bicubic_impl<0, 4, true>(P, out_idx, in_mode, stream);
}
} // namespace lrp

#if defined(LRP_WAVE_STAMPS)
extern "C" void lrp_debug_read_wave_stamps(unsigned long long *out, int n_waves) { // (this unit's instantiations: RGBA, coordinates from the geometry cache)
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lrp::g_wave_stamps), sizeof(unsigned long long) * 3 * (size_t)n_waves);
}
#endif
