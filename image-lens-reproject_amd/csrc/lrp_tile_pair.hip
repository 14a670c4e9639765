// lrp_tile_pair.hip — the pair kernel (lrp_pair_kernel.h): alias pairs of in-view blocks, two wavefronts per window; RGB, RGBA, RGBAZ.
#include "lrp_pair_kernel.h"

namespace lrp {
hipError_t launch_pair_kernel(const KParams &P, hipStream_t stream) { return launch_pair_bicubic(P, stream); }
} // namespace lrp
