// lrp_image_io.h — image codecs of the CLI (row f3 of SURVEY.md §8f): the in-memory
// conventions of reference src/image_formats.cpp:144-345, re-implemented on what
// this image ships (libpng + zlib; OpenEXR / lodepng are absent).
//
//   PNG  read : 8-bit RGBA decode (16-bit samples reduced to their high byte, like
//               lodepng), alpha dropped, v = pow(p / 255, 2.2), C = 3   (:174-204)
//        write: uint8(255.9 * pow(clamp(v, 0, 1), 1 / 2.2)), RGBA8, alpha 255
//               unless C == 4                                            (:144-172)
//   EXR  read : every channel as HALF widened to float (FLOAT channels go through
//               half first, as OpenEXR's HALF frame-buffer slices do), layout
//               RGB / RGBA / RGBZ / RGBAZ from the presence of A and Z  (:208-303)
//        write: HALF channels R, G, B, A, Z (first C of them), ZIP       (:305-345)
//   JPEG read : three components, v = pow(p / 255, 2.2)                  (:26-77)
//        write: uint8(255.9 * pow(clamp(v, 0, 1), 1 / 2.2)), quality 95  (:79-142)
//   Scanline single-part EXR with NO / ZIPS / ZIP compression only; JPEG through the image's IJG
//   libjpeg 9 runtime, opened with dlopen (cli/lrp_jpeg.cpp).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace lrp_io {

struct Frame {
  int width = 0, height = 0, channels = 0;
  int data_layout = 0; // reproject::DataLayout numbering: RGB 0, RGBA 1, RGBZ 2, RGBAZ 3
  std::vector<float> data;
};

Frame read_png(const std::string &path);
void save_png(const Frame &f, const std::string &path);
Frame read_exr(const std::string &path);
void save_exr(const Frame &f, const std::string &path);
Frame read_jpeg(const std::string &path);
void save_jpeg(const Frame &f, const std::string &path);

// binary16 <-> binary32, round to nearest even (what OpenEXR's `half` does)
uint16_t float_to_half(float f);
float half_to_float(uint16_t h);

} // namespace lrp_io
