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

#include <cstddef>
#include <cstdint>
#include <utility>
#include <stdexcept>
#include <string>
#include <vector>

namespace lrp_io {

struct Frame {
  int width = 0, height = 0, channels = 0;
  int data_layout = 0; // reproject::DataLayout numbering: RGB 0, RGBA 1, RGBZ 2, RGBAZ 3
  std::vector<float> data;
};

// A frame in its FILE format — what crosses PCIe when the conversions run on the device
// (lrp_context_submit_packed): RGBA8 / RGB8 samples of a PNG / JPEG (LRP_PIXEL_U8_GAMMA) or
// interleaved binary16 samples of an EXR (LRP_PIXEL_F16).  `channels` is what the hot path
// sees (PNG: 3 of the 4 decoded samples), `packed_channels` what the buffer holds per pixel.
// The buffer comes from `alloc` (the CLI passes page-locked memory) and is released with it.
struct Allocator {
  void *(*alloc)(size_t bytes);
  void (*release)(void *ptr);
};
Allocator heap_allocator();

struct Packed {
  int width = 0, height = 0, channels = 0, packed_channels = 0;
  int data_layout = 0;
  int format = 0; // lrp_pixel_format numbering: 1 binary16, 2 8-bit gamma
  uint8_t *bytes = nullptr;
  size_t size = 0;
  Allocator allocator{nullptr, nullptr};
  Packed() = default;
  Packed(const Packed &) = delete;
  Packed &operator=(const Packed &) = delete;
  Packed(Packed &&o) noexcept { *this = std::move(o); }
  Packed &operator=(Packed &&o) noexcept;
  ~Packed();
  void allocate(const Allocator &a, size_t n);
};

// by extension: .exr -> binary16, .png -> RGBA8, .jpg / .jpeg -> RGB8
Packed read_packed(const std::string &path, const Allocator &alloc);
Packed read_png_packed(const std::string &path, const Allocator &alloc);
Packed read_exr_packed(const std::string &path, const Allocator &alloc);
Packed read_jpeg_packed(const std::string &path, const Allocator &alloc);
// RGBA8 with width x height x 4 bytes / binary16 with width x height x channels samples
void save_png_rgba8(const uint8_t *rgba, int width, int height, const std::string &path);
void save_exr_half(const uint16_t *half_pixels, int width, int height, int channels, const std::string &path);
// the host-side conversions between the two representations (the device kernels' twins)
Frame unpack(const Packed &p);

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
