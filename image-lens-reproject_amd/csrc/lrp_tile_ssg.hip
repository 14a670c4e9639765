// lrp_tile_ssg.hip — nearest / bilinear with num_samples 2-4 from an entry of sub-samples (lrp_ss_gather_kernel.h).
#include "lrp_ss_gather_kernel.h"

namespace lrp {
hipError_t launch_ss_gather(const KParams &P, int interpolation, int in_mode, hipStream_t stream) {
  return launch_ss_gather_impl(P, interpolation, in_mode, stream);
}
} // namespace lrp
