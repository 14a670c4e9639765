// lrp_tile_bl.hip — bilinear instantiations of the tile kernel (lrp_kernel_v2.h).
#include "lrp_kernel_v2.h"

namespace lrp {
hipError_t launch_tile_bilinear(const KParams &P, int out_idx, int in_mode, hipStream_t stream) {
  return launch_tile_interp<1>(P, out_idx, in_mode, stream);
}
} // namespace lrp
