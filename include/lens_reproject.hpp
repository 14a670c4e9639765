// lens_reproject.hpp — C++ face of liblrp_hip.so with the reference's operator
// interface for the hot path:
//
//     reproject::reproject(const Image *in, Image *out, int num_samples,
//                          Interpolation interpolation, const float *rotation_matrix);
//     reproject::post_process(const Image *img, float exposure, float reinhard);
//     reproject::test_conversion_math();
//
// i.e. the declarations of reference src/reproject.hpp:7-27 and the lens types of
// reference src/config.hpp:7-37 (same names, same enumerator order, same struct
// layout: sizeof(LensInfo) == 28, sizeof(Image) == 56), implemented by forwarding
// to the C ABI of include/lrp.h.  A project that already has the reference's own
// headers keeps them and compiles integration/reproject_hip.cpp instead of this
// file; this header is for code that has neither.
//
// Behaviour kept from the reference (src/reproject.cpp:364-366,395-397,415-417):
// an unsupported output lens, input lens or interpolation prints the reference's
// message to stdout and calls std::exit(1).  Every other failure (no GPU, HIP
// error, out of memory, channel mismatch) throws std::runtime_error, which the
// reference's per-file worker catches and reports (src/main.cpp:617-619).
// There is no CPU fallback.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "lrp.h"

namespace reproject {

enum LensType { RECTILINEAR, FISHEYE_EQUIDISTANT, FISHEYE_EQUISOLID, FISHEYE_STEREOGRAPHIC, EQUIRECTANGULAR };

struct LensInfo {
  LensType type;
  union {
    struct {
      float focal_length;
    } rectilinear;
    struct {
      float fov;
    } fisheye_equidistant;
    struct {
      float focal_length;
      float fov;
    } fisheye_equisolid;
    struct {
      float latitude_min;
      float latitude_max;
      float longitude_min;
      float longitude_max;
    } equirectangular;
  };
  float sensor_width;
  float sensor_height;
};

enum DataLayout { RGB, RGBA, RGBZ, RGBAZ };

struct Image {
  LensInfo lens;
  int width, height, channels;
  float *data;
  DataLayout data_layout;
};

enum Interpolation { NEAREST, BILINEAR, BICUBIC };

static_assert(sizeof(LensInfo) == sizeof(lrp_lens) && sizeof(LensInfo) == 28, "LensInfo layout");
static_assert(sizeof(Image) == sizeof(lrp_image) && sizeof(Image) == 56, "Image layout");

#ifdef LRP_DECLARATIONS_ONLY
// Declarations only — exactly what reference src/reproject.hpp:22-27 declares;
// the definitions come from integration/reproject_hip.cpp.
void reproject(const Image *in, Image *out, int num_samples, Interpolation interpolation, const float *rotation_matrix);
void post_process(const Image *img, float exposure, float reinhard);
void test_conversion_math();
#else

namespace detail {
inline lrp_image to_c(const Image *im) {
  lrp_image c;
  std::memcpy(&c, im, sizeof(c)); // identical layout (static_asserts above, tests/test_abi.py)
  return c;
}
// GPU used by the calling thread: LRP_DEVICE in the environment, else 0.  A
// scheduler that spreads pool threads over GPUs sets lrp_thread_device instead.
inline int &thread_device() {
  static thread_local int dev = [] {
    const char *e = std::getenv("LRP_DEVICE");
    return e ? std::atoi(e) : 0;
  }();
  return dev;
}
[[noreturn]] inline void die(const char *msg) {
  std::printf("%s\n", msg);
  std::exit(1);
}
inline void check(int status) {
  switch (status) {
  case LRP_OK: return;
  case LRP_ERR_OUTPUT_LENS: die("Output lens type not supported.");
  case LRP_ERR_INPUT_LENS: die("Input lens type not supported.");
  case LRP_ERR_INTERPOLATION: die("Interpolation method not supported.");
  default: {
    std::string what = lrp_strerror(status);
    const char *detail = lrp_last_error();
    if (detail && *detail) what += std::string(": ") + detail;
    throw std::runtime_error(what);
  }
  }
}
} // namespace detail

/// Select the GPU the calling thread's reproject()/post_process() calls use.
inline void set_thread_device(int device) { detail::thread_device() = device; }

inline void reproject(const Image *in, Image *out, int num_samples, Interpolation interpolation,
                      const float *rotation_matrix) {
  const lrp_image cin = detail::to_c(in);
  lrp_image cout = detail::to_c(out);
  detail::check(lrp_reproject(&cin, &cout, num_samples, (int)interpolation, rotation_matrix, nullptr,
                              detail::thread_device()));
}

/// reproject() followed by post_process() in one kernel (the CLI sequence of
/// reference src/main.cpp:597-603 when exposure != 1 || reinhard != 1).
inline void reproject_and_post_process(const Image *in, Image *out, int num_samples, Interpolation interpolation,
                                       const float *rotation_matrix, float exposure, float reinhard) {
  const lrp_image cin = detail::to_c(in);
  lrp_image cout = detail::to_c(out);
  const lrp_post post{exposure, reinhard};
  detail::check(lrp_reproject(&cin, &cout, num_samples, (int)interpolation, rotation_matrix, &post,
                              detail::thread_device()));
}

inline void post_process(const Image *img, float exposure, float reinhard) {
  lrp_image c = detail::to_c(img);
  detail::check(lrp_post_process(&c, exposure, reinhard, detail::thread_device()));
}

/// An empty function in the reference (src/reproject.cpp:467), called once at
/// program start (src/main.cpp:147).
inline void test_conversion_math() {}

#endif // LRP_DECLARATIONS_ONLY

} // namespace reproject
