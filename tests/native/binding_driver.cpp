// Drives the C++ boundary the way the reference's worker does (src/main.cpp:576-603).
// usage: binding_driver <case>
//   out_lens | in_lens | interp : unsupported dispatch -> reference message + exit(1)
//   nodevice : a valid call; without a GPU the binding must throw (no CPU fallback)
//   run [file] : a valid call on the GPU; writes the raw output floats to `file` (the test compares
//                every bit with the oracle) and prints a checksum
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "reproject.hpp"
#include "lrp.h"

int main(int argc, char **argv) {
  const char *mode = argc > 1 ? argv[1] : "run";
  reproject::test_conversion_math();
  const int w = 64, h = 32, c = 4;
  std::vector<float> src((size_t)w * h * c), dst((size_t)w * h * c, -1.0f);
  for (size_t i = 0; i < src.size(); ++i) src[i] = (float)(((uint32_t)i * 2654435761u) >> 21) / 2048.0f;
  reproject::Image in{}, out{};
  lrp_lens tmp;
  lrp_lens_equirectangular_full(&tmp);
  std::memcpy(&in.lens, &tmp, sizeof(tmp));
  lrp_lens_rectilinear(&tmp, 18.0f, 36.0f, (float)w, (float)h);
  std::memcpy(&out.lens, &tmp, sizeof(tmp));
  in.width = out.width = w;
  in.height = out.height = h;
  in.channels = out.channels = c;
  in.data = src.data();
  out.data = dst.data();
  in.data_layout = out.data_layout = reproject::RGBA;
  reproject::Interpolation interp = reproject::BICUBIC;
  if (!std::strcmp(mode, "out_lens")) out.lens.type = reproject::FISHEYE_EQUISOLID;
  if (!std::strcmp(mode, "in_lens")) in.lens.type = reproject::FISHEYE_STEREOGRAPHIC;
  if (!std::strcmp(mode, "interp")) interp = (reproject::Interpolation)7;
  float rot[9];
  lrp_rotation_matrix(0.5f, -0.25f, 0.1f, rot);
  try {
    reproject::reproject(&in, &out, 1, interp, rot);
    reproject::post_process(&out, 2.0f, 4.0f);
  } catch (const std::exception &e) { // the reference worker's catch (src/main.cpp:617-619)
    std::printf("Error: %s\n", e.what());
    return 3;
  }
  if (argc > 2) {
    std::FILE *f = std::fopen(argv[2], "wb");
    if (!f || std::fwrite(dst.data(), sizeof(float), dst.size(), f) != dst.size()) {
      std::printf("Error: cannot write %s\n", argv[2]);
      return 4;
    }
    std::fclose(f);
  }
  double sum = 0;
  for (float v : dst) sum += v;
  std::printf("ok sum=%.9g\n", sum);
  return 0;
}
