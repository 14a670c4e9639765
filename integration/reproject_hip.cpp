// reproject_hip.cpp — the binding a maintainer of IDLabMedia/image-lens-reproject
// adds to run the hot path on an MI355X: compile THIS file instead of
// src/reproject.cpp and link liblrp_hip.so (INTEGRATION.md has the CMake lines).
//
// It includes the project's own "reproject.hpp" (reference src/reproject.hpp,
// which pulls in src/config.hpp) and defines the three symbols that header
// declares (src/reproject.hpp:22-27) by forwarding to the C ABI of include/lrp.h.
// Nothing else in the project changes: src/main.cpp:597-603 keeps calling
//     reproject::reproject(&input, &output, num_samples, interpolation, rotation_matrix);
//     reproject::post_process(&output, exposure, reinhard);
//
// When built outside the reference tree (tests/test_cxx_binding.py) the include
// below resolves to tests/native/reproject.hpp, which is include/lens_reproject.hpp
// in its declarations-only form (the same declarations as the reference header).
#include "reproject.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "lrp.h"

namespace reproject {

static_assert(sizeof(LensInfo) == sizeof(lrp_lens), "LensInfo and lrp_lens must have the same layout");
static_assert(sizeof(Image) == sizeof(lrp_image), "Image and lrp_image must have the same layout");

namespace {
int device_of_this_thread() {
  static thread_local int dev = [] {
    const char *e = std::getenv("LRP_DEVICE");
    return e ? std::atoi(e) : 0;
  }();
  return dev;
}
lrp_image to_c(const Image *im) {
  lrp_image c;
  std::memcpy(&c, im, sizeof(c));
  return c;
}
void check(int status) {
  const char *msg = nullptr;
  switch (status) {
  case LRP_OK: return;
  case LRP_ERR_OUTPUT_LENS: msg = "Output lens type not supported."; break;
  case LRP_ERR_INPUT_LENS: msg = "Input lens type not supported."; break;
  case LRP_ERR_INTERPOLATION: msg = "Interpolation method not supported."; break;
  default: break;
  }
  if (msg) { // same line + exit code as src/reproject.cpp:365-366,396-397,416-417
    std::printf("%s\n", msg);
    std::exit(1);
  }
  throw std::runtime_error(std::string(lrp_strerror(status)) + ": " + lrp_last_error());
}
} // namespace

void reproject(const Image *in, Image *out, int num_samples, Interpolation interpolation,
               const float *rotation_matrix) {
  const lrp_image cin = to_c(in);
  lrp_image cout = to_c(out);
  check(lrp_reproject(&cin, &cout, num_samples, (int)interpolation, rotation_matrix, nullptr, device_of_this_thread()));
}

void post_process(const Image *img, float exposure, float reinhard) {
  lrp_image c = to_c(img);
  check(lrp_post_process(&c, exposure, reinhard, device_of_this_thread()));
}

void test_conversion_math() {}

} // namespace reproject
