// lrp_kernels_bc.hip — bicubic instantiations of the reprojection kernel
// (one translation unit per interpolation mode; see lrp_kernel_impl.h).
#include "lrp_kernel_impl.h"

namespace lrp {
hipError_t launch_bicubic(const KParams &P, int out_idx, int in_mode, hipStream_t stream) {
  return launch_interp<2>(P, out_idx, in_mode, stream);
}
} // namespace lrp
