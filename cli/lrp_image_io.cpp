// lrp_image_io.cpp — see lrp_image_io.h.
#include "lrp_image_io.h"

#include "lrp_half.h"

#include <png.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

namespace lrp_io {

// ---------------------------------------------------------------- half <-> float
// (one implementation for the host codecs and the device kernels: include/lrp_half.h)
float half_to_float(uint16_t h) {
  const uint32_t bits = lrp_half_to_float_bits(h);
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

uint16_t float_to_half(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  return lrp_float_bits_to_half(x);
}

// ---------------------------------------------------------------- packed frames
Allocator heap_allocator() { return Allocator{[](size_t n) { return std::malloc(n ? n : 1); }, [](void *p) { std::free(p); }}; }

Packed &Packed::operator=(Packed &&o) noexcept {
  if (this != &o) {
    if (bytes && allocator.release) allocator.release(bytes);
    width = o.width;
    height = o.height;
    channels = o.channels;
    packed_channels = o.packed_channels;
    data_layout = o.data_layout;
    format = o.format;
    bytes = o.bytes;
    size = o.size;
    allocator = o.allocator;
    o.bytes = nullptr;
    o.size = 0;
  }
  return *this;
}
Packed::~Packed() {
  if (bytes && allocator.release) allocator.release(bytes);
}
void Packed::allocate(const Allocator &a, size_t n) {
  if (bytes && allocator.release) allocator.release(bytes);
  allocator = a;
  bytes = static_cast<uint8_t *>(a.alloc(n));
  size = n;
  if (!bytes) throw std::runtime_error("out of memory for an image buffer");
}

Frame unpack(const Packed &p) {
  Frame f;
  f.width = p.width;
  f.height = p.height;
  f.channels = p.channels;
  f.data_layout = p.data_layout;
  const size_t n = (size_t)p.width * p.height;
  f.data.resize(n * p.channels);
  if (p.format == 1) {
    const uint16_t *h = reinterpret_cast<const uint16_t *>(p.bytes);
    for (size_t i = 0; i < n; ++i)
      for (int c = 0; c < p.channels; ++c) f.data[i * p.channels + c] = half_to_float(h[i * p.packed_channels + c]);
  } else {
    // 256-entry table of the reference's per-sample conversion pow(float(p) / 255.0f, 2.2f)
    // (src/image_formats.cpp:64-66, 196-198)
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = std::pow(float(i) / 255.0f, 2.2f);
    for (size_t i = 0; i < n; ++i)
      for (int c = 0; c < p.channels; ++c) f.data[i * p.channels + c] = lut[p.bytes[i * p.packed_channels + c]];
  }
  return f;
}

Packed read_packed(const std::string &path, const Allocator &alloc) {
  const size_t dot = path.rfind('.');
  const std::string ext = dot == std::string::npos ? std::string() : path.substr(dot);
  if (ext == ".exr") return read_exr_packed(path, alloc);
  if (ext == ".png") return read_png_packed(path, alloc);
  if (ext == ".jpg" || ext == ".jpeg") return read_jpeg_packed(path, alloc);
  throw std::runtime_error("Input format not supported: " + ext);
}

// ---------------------------------------------------------------- PNG
namespace {
// to 8-bit RGBA the way lodepng's default decode does: 16-bit samples keep their
// high byte, palettes and grey levels expand, tRNS becomes alpha, no gamma handling
bool png_header(png_structp png, png_infop info, FILE *fp, unsigned *w, unsigned *h) {
  if (setjmp(png_jmpbuf(png))) return false;
  png_init_io(png, fp);
  png_read_info(png, info);
  *w = png_get_image_width(png, info);
  *h = png_get_image_height(png, info);
  const int color = png_get_color_type(png, info), depth = png_get_bit_depth(png, info);
  if (depth == 16) png_set_strip_16(png);
  if (color == PNG_COLOR_TYPE_PALETTE) png_set_palette_to_rgb(png);
  if (color == PNG_COLOR_TYPE_GRAY && depth < 8) png_set_expand_gray_1_2_4_to_8(png);
  if (png_get_valid(png, info, PNG_INFO_tRNS)) png_set_tRNS_to_alpha(png);
  if (color == PNG_COLOR_TYPE_GRAY || color == PNG_COLOR_TYPE_GRAY_ALPHA) png_set_gray_to_rgb(png);
  png_set_filler(png, 0xff, PNG_FILLER_AFTER);
  png_set_interlace_handling(png);
  png_read_update_info(png, info);
  return true;
}
bool png_rows(png_structp png, png_bytep *rows) {
  if (setjmp(png_jmpbuf(png))) return false;
  png_read_image(png, rows);
  png_read_end(png, nullptr);
  return true;
}
} // namespace

Packed read_png_packed(const std::string &path, const Allocator &alloc) {
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) throw std::runtime_error("cannot open " + path);
  png_structp png = png_create_read_struct(PNG_LIBPNG_VER_STRING, nullptr, nullptr, nullptr);
  png_infop info = png ? png_create_info_struct(png) : nullptr;
  if (!png || !info) {
    if (png) png_destroy_read_struct(&png, nullptr, nullptr);
    std::fclose(fp);
    throw std::runtime_error("libpng initialisation failed");
  }
  // libpng reports errors by longjmp: the two frames that call into it (png_header / png_rows above) hold
  // plain data only, the buffers live here and are released by ordinary unwinding.
  unsigned w = 0, h = 0;
  if (!png_header(png, info, fp, &w, &h) || w == 0 || h == 0 || w > 65535u || h > 65535u) {
    png_destroy_read_struct(&png, &info, nullptr);
    std::fclose(fp);
    throw std::runtime_error("cannot decode PNG " + path);
  }
  Packed p;
  p.width = (int)w;
  p.height = (int)h;
  p.channels = 3; // alpha is decoded and dropped (src/image_formats.cpp:186-198)
  p.packed_channels = 4;
  p.data_layout = 0;
  p.format = 2;
  bool ok = false;
  try {
    p.allocate(alloc, (size_t)w * h * 4);
    std::vector<png_bytep> rows(h);
    for (unsigned y = 0; y < h; ++y) rows[y] = p.bytes + (size_t)y * w * 4;
    ok = png_rows(png, rows.data());
  } catch (...) {
    png_destroy_read_struct(&png, &info, nullptr);
    std::fclose(fp);
    throw;
  }
  png_destroy_read_struct(&png, &info, nullptr);
  std::fclose(fp);
  if (!ok) throw std::runtime_error("cannot decode PNG " + path);
  return p;
}

Frame read_png(const std::string &path) { return unpack(read_png_packed(path, heap_allocator())); }

void save_png(const Frame &f, const std::string &path) {
  std::vector<uint8_t> buf((size_t)f.width * f.height * 4);
  for (size_t i = 0, n = (size_t)f.width * f.height; i < n; ++i) {
    for (int c = 0; c < f.channels && c < 4; ++c) { // src/image_formats.cpp:151-159 (a 5th channel would overrun there)
      float s = f.data[i * f.channels + c];
      s = std::max(0.0f, std::min(1.0f, s));
      s = std::pow(s, 1.0f / 2.2f);
      buf[i * 4 + c] = (uint8_t)(255.9f * s);
    }
    if (f.channels != 4) buf[i * 4 + 3] = 255;
  }
  save_png_rgba8(buf.data(), f.width, f.height, path);
}

namespace {
bool png_write_all(png_structp png, png_infop info, FILE *fp, const uint8_t *rgba, int width, int height) {
  if (setjmp(png_jmpbuf(png))) return false;
  png_init_io(png, fp);
  png_set_IHDR(png, info, (png_uint_32)width, (png_uint_32)height, 8, PNG_COLOR_TYPE_RGBA, PNG_INTERLACE_NONE,
               PNG_COMPRESSION_TYPE_DEFAULT, PNG_FILTER_TYPE_DEFAULT);
  png_write_info(png, info);
  for (int y = 0; y < height; ++y) png_write_row(png, const_cast<png_bytep>(rgba + (size_t)y * width * 4));
  png_write_end(png, nullptr);
  return true;
}
} // namespace

void save_png_rgba8(const uint8_t *rgba, int width, int height, const std::string &path) {
  FILE *fp = std::fopen(path.c_str(), "wb");
  if (!fp) throw std::runtime_error("cannot write " + path);
  png_structp png = png_create_write_struct(PNG_LIBPNG_VER_STRING, nullptr, nullptr, nullptr);
  png_infop info = png ? png_create_info_struct(png) : nullptr;
  const bool ok = png && info && png_write_all(png, info, fp, rgba, width, height);
  if (png) png_destroy_write_struct(&png, &info);
  std::fclose(fp);
  if (!ok) throw std::runtime_error("cannot encode PNG " + path);
}

// ---------------------------------------------------------------- EXR
namespace {

struct Reader {
  const std::vector<uint8_t> &b;
  size_t pos = 0;
  explicit Reader(const std::vector<uint8_t> &bytes) : b(bytes) {}
  // (written so that neither a huge n nor a pos beyond the end can wrap the comparison)
  void need(size_t n) const {
    if (n > b.size() || pos > b.size() - n) throw std::runtime_error("truncated EXR file");
  }
  // a size / count field of the file: 32-bit signed on disk, never negative
  size_t get_size() {
    const int32_t v = get<int32_t>();
    if (v < 0) throw std::runtime_error("corrupt EXR file (negative size)");
    return (size_t)v;
  }
  void skip(size_t n) {
    need(n);
    pos += n;
  }
  template <typename T> T get() {
    need(sizeof(T));
    T v;
    std::memcpy(&v, b.data() + pos, sizeof(T));
    pos += sizeof(T);
    return v;
  }
  std::string str() {
    std::string s;
    for (;;) {
      need(1);
      const char c = (char)b[pos++];
      if (!c) break;
      s.push_back(c);
    }
    return s;
  }
};

struct Channel {
  std::string name;
  int32_t type; // 0 UINT, 1 HALF, 2 FLOAT
};

std::vector<uint8_t> read_file(const std::string &path) {
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) throw std::runtime_error("cannot open " + path);
  std::fseek(fp, 0, SEEK_END);
  const long n = std::ftell(fp);
  std::fseek(fp, 0, SEEK_SET);
  std::vector<uint8_t> bytes((size_t)std::max(0l, n));
  if (n > 0 && std::fread(bytes.data(), 1, (size_t)n, fp) != (size_t)n) {
    std::fclose(fp);
    throw std::runtime_error("cannot read " + path);
  }
  std::fclose(fp);
  return bytes;
}

// OpenEXR's zip post-processing: byte predictor + split into even / odd halves
void unpredict_and_interleave(const std::vector<uint8_t> &in, uint8_t *out) {
  std::vector<uint8_t> t(in);
  for (size_t i = 1; i < t.size(); ++i) t[i] = (uint8_t)(t[i - 1] + t[i] - 128);
  const size_t half = (t.size() + 1) / 2;
  for (size_t i = 0, o = 0; o < t.size(); ++i) {
    out[o++] = t[i];
    if (o < t.size()) out[o++] = t[half + i];
  }
}
std::vector<uint8_t> split_and_predict(const uint8_t *raw, size_t n) {
  std::vector<uint8_t> t(n);
  const size_t half = (n + 1) / 2;
  for (size_t i = 0, a = 0, b = half; i < n; ++i) {
    if (i & 1)
      t[b++] = raw[i];
    else
      t[a++] = raw[i];
  }
  for (size_t i = n; i-- > 1;) t[i] = (uint8_t)(t[i] - t[i - 1] + 128);
  return t;
}

} // namespace

Packed read_exr_packed(const std::string &path, const Allocator &alloc) {
  const std::vector<uint8_t> bytes = read_file(path);
  Reader r(bytes);
  if (r.get<uint32_t>() != 20000630u) throw std::runtime_error(path + ": not an OpenEXR file");
  const uint32_t version = r.get<uint32_t>();
  if ((version & 0xffu) != 2 || (version & 0x1a00u)) // tiled 0x200, deep 0x800, multipart 0x1000
    throw std::runtime_error(path + ": only single-part scanline OpenEXR files are supported");
  std::vector<Channel> channels;
  int compression = -1, line_order = 0;
  int32_t dw[4] = {0, 0, -1, -1};
  for (;;) {
    const std::string name = r.str();
    if (name.empty()) break;
    const std::string type = r.str();
    const size_t size = r.get_size();
    r.need(size);
    const size_t end = r.pos + size;
    if (name == "channels") {
      for (;;) {
        const std::string cn = r.str();
        if (cn.empty()) break;
        Channel c{cn, r.get<int32_t>()};
        r.skip(4); // pLinear + reserved
        const int32_t xs = r.get<int32_t>(), ys = r.get<int32_t>();
        if (xs != 1 || ys != 1) throw std::runtime_error(path + ": sub-sampled channels are not supported");
        channels.push_back(c);
      }
    } else if (name == "compression") {
      compression = r.get<uint8_t>();
    } else if (name == "dataWindow") {
      for (int i = 0; i < 4; ++i) dw[i] = r.get<int32_t>();
    } else if (name == "lineOrder") {
      line_order = r.get<uint8_t>();
    }
    r.pos = end;
  }
  (void)line_order; // every chunk carries its own y
  if (channels.empty() || dw[2] < dw[0] || dw[3] < dw[1]) throw std::runtime_error(path + ": incomplete OpenEXR header");
  // sizes from the file are bounded before anything is allocated or multiplied (64-bit, no int overflow)
  const int64_t width64 = (int64_t)dw[2] - dw[0] + 1, height64 = (int64_t)dw[3] - dw[1] + 1;
  if (width64 > 65535 || height64 > 65535 || channels.size() > 64)
    throw std::runtime_error(path + ": OpenEXR data window or channel list too large");
  int lines_per_block;
  if (compression == 0 || compression == 2)
    lines_per_block = 1;
  else if (compression == 3)
    lines_per_block = 16;
  else
    throw std::runtime_error(path + ": only NO / ZIPS / ZIP compression is supported (found type " +
                             std::to_string(compression) + ")");
  Packed f;
  f.width = (int)width64;
  f.height = (int)height64;
  f.channels = f.packed_channels = (int)channels.size();
  f.format = 1;
  bool has_a = false, has_z = false;
  for (const Channel &c : channels) {
    has_a |= c.name == "A";
    has_z |= c.name == "Z";
  }
  f.data_layout = has_a && has_z ? 3 : (has_a ? 1 : (has_z ? 2 : 0)); // src/image_formats.cpp:232-240
  // destination channel of each file channel, src/image_formats.cpp:263-283; channels that
  // map nowhere in the reference (dstC = -1000, out of bounds there) are rejected here
  std::vector<int> dst(channels.size(), 0);
  for (size_t i = 0; i < channels.size(); ++i) {
    const std::string &n = channels[i].name;
    int d = 0;
    if (n == "R") d = 0;
    if (n == "G") d = 1;
    if (n == "B") d = 2;
    if (f.data_layout == 1 && n == "A") d = 3;
    if (f.data_layout == 2 && n == "Z") d = 3;
    if (f.data_layout == 3 && n == "A") d = 3;
    if (f.data_layout == 3 && n == "Z") d = 4;
    if (d >= f.channels) throw std::runtime_error(path + ": unexpected channel set");
    dst[i] = d;
  }
  size_t line_bytes = 0;
  for (const Channel &c : channels) {
    if (c.type != 1 && c.type != 2) throw std::runtime_error(path + ": UINT channels are not supported");
    line_bytes += (size_t)f.width * (c.type == 1 ? 2 : 4);
  }
  f.allocate(alloc, (size_t)f.width * f.height * f.channels * 2);
  std::memset(f.bytes, 0, f.size);
  uint16_t *const pixels = reinterpret_cast<uint16_t *>(f.bytes);
  const int n_blocks = (f.height + lines_per_block - 1) / lines_per_block;
  std::vector<uint64_t> offsets((size_t)n_blocks);
  for (auto &o : offsets) o = r.get<uint64_t>();
  std::vector<uint8_t> raw;
  for (int bi = 0; bi < n_blocks; ++bi) {
    Reader c(bytes);
    if (offsets[(size_t)bi] > bytes.size()) throw std::runtime_error(path + ": scanline block offset beyond the file");
    c.pos = (size_t)offsets[(size_t)bi];
    const int32_t y0 = c.get<int32_t>();
    const size_t size = c.get_size();
    c.need(size);
    const int64_t row0_64 = (int64_t)y0 - dw[1];
    if (row0_64 < 0 || row0_64 >= f.height) throw std::runtime_error(path + ": scanline block outside the data window");
    const int row0 = (int)row0_64;
    const int n_lines = std::min(lines_per_block, f.height - row0);
    const size_t want = line_bytes * (size_t)n_lines;
    raw.resize(want);
    if (compression == 0 || size == want) {
      if (size != want) throw std::runtime_error(path + ": bad uncompressed block size");
      std::memcpy(raw.data(), bytes.data() + c.pos, want);
    } else {
      std::vector<uint8_t> tmp(want);
      uLongf got = (uLongf)want;
      if (uncompress(tmp.data(), &got, bytes.data() + c.pos, (uLong)size) != Z_OK || got != want)
        throw std::runtime_error(path + ": zlib error in scanline block");
      unpredict_and_interleave(tmp, raw.data());
    }
    const uint8_t *p = raw.data();
    for (int l = 0; l < n_lines; ++l) {
      const int y = row0 + l;
      for (size_t ci = 0; ci < channels.size(); ++ci) {
        uint16_t *out = pixels + ((size_t)y * f.width) * f.channels + dst[ci];
        if (channels[ci].type == 1) {
          for (int x = 0; x < f.width; ++x, p += 2) std::memcpy(&out[(size_t)x * f.channels], p, 2);
        } else {
          for (int x = 0; x < f.width; ++x, p += 4) {
            float v;
            std::memcpy(&v, p, 4);
            out[(size_t)x * f.channels] = float_to_half(v); // read through a HALF slice
          }
        }
      }
    }
  }
  return f;
}

Frame read_exr(const std::string &path) { return unpack(read_exr_packed(path, heap_allocator())); }

void save_exr(const Frame &f, const std::string &path) {
  if (f.channels > 5) throw std::runtime_error("cannot save exr with more than 5 channels."); // :312
  std::vector<uint16_t> h(f.data.size());
  for (size_t i = 0; i < h.size(); ++i) h[i] = float_to_half(f.data[i]);
  save_exr_half(h.data(), f.width, f.height, f.channels, path);
}

void save_exr_half(const uint16_t *half_pixels, int width, int height, int n_channels, const std::string &path) {
  static const char *kNames[5] = {"R", "G", "B", "A", "Z"};
  if (n_channels > 5) throw std::runtime_error("cannot save exr with more than 5 channels."); // :312
  struct { int width, height, channels; } f{width, height, n_channels};
  // file order = alphabetical channel order
  std::vector<int> order;
  for (int c = 0; c < f.channels; ++c) order.push_back(c);
  std::sort(order.begin(), order.end(), [](int a, int b) { return std::strcmp(kNames[a], kNames[b]) < 0; });
  std::vector<uint8_t> head;
  auto put = [&](const void *p, size_t n) { head.insert(head.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
  auto put_str = [&](const char *s) { put(s, std::strlen(s) + 1); };
  auto put_i32 = [&](int32_t v) { put(&v, 4); };
  auto put_f32 = [&](float v) { put(&v, 4); };
  auto attr = [&](const char *name, const char *type, int32_t size) {
    put_str(name);
    put_str(type);
    put_i32(size);
  };
  const uint32_t magic = 20000630u, version = 2;
  put(&magic, 4);
  put(&version, 4);
  attr("channels", "chlist", (int32_t)(f.channels * (2 + 16) + 1));
  for (int c : order) {
    put_str(kNames[c]);
    put_i32(1); // HALF
    const uint8_t lin[4] = {0, 0, 0, 0};
    put(lin, 4);
    put_i32(1);
    put_i32(1);
  }
  head.push_back(0);
  attr("compression", "compression", 1);
  head.push_back(3); // ZIP, 16 scanlines per block
  const int32_t win[4] = {0, 0, f.width - 1, f.height - 1};
  attr("dataWindow", "box2i", 16);
  put(win, 16);
  attr("displayWindow", "box2i", 16);
  put(win, 16);
  attr("lineOrder", "lineOrder", 1);
  head.push_back(0);
  attr("pixelAspectRatio", "float", 4);
  put_f32(1.0f);
  attr("screenWindowCenter", "v2f", 8);
  put_f32(0.0f);
  put_f32(0.0f);
  attr("screenWindowWidth", "float", 4);
  put_f32(1.0f);
  head.push_back(0);

  const int lines_per_block = 16;
  const int n_blocks = (f.height + lines_per_block - 1) / lines_per_block;
  const size_t line_bytes = (size_t)f.width * f.channels * 2;
  std::vector<std::vector<uint8_t>> blocks((size_t)n_blocks);
  std::vector<uint8_t> raw;
  for (int bi = 0; bi < n_blocks; ++bi) {
    const int row0 = bi * lines_per_block, n_lines = std::min(lines_per_block, f.height - row0);
    raw.resize(line_bytes * (size_t)n_lines);
    uint8_t *p = raw.data();
    for (int l = 0; l < n_lines; ++l)
      for (int c : order)
        for (int x = 0; x < f.width; ++x, p += 2) {
          std::memcpy(p, &half_pixels[((size_t)(row0 + l) * f.width + x) * f.channels + c], 2);
        }
    const std::vector<uint8_t> t = split_and_predict(raw.data(), raw.size());
    uLongf bound = compressBound((uLong)t.size());
    std::vector<uint8_t> z(bound);
    if (compress2(z.data(), &bound, t.data(), (uLong)t.size(), 9) != Z_OK) throw std::runtime_error("zlib error"); // level 9, :330
    std::vector<uint8_t> &blk = blocks[(size_t)bi];
    const bool store_raw = bound >= raw.size();
    const int32_t y = row0, size = (int32_t)(store_raw ? raw.size() : bound);
    blk.resize(8 + (size_t)size);
    std::memcpy(blk.data(), &y, 4);
    std::memcpy(blk.data() + 4, &size, 4);
    std::memcpy(blk.data() + 8, store_raw ? raw.data() : z.data(), (size_t)size);
  }
  FILE *fp = std::fopen(path.c_str(), "wb");
  if (!fp) throw std::runtime_error("cannot write " + path);
  std::fwrite(head.data(), 1, head.size(), fp);
  uint64_t off = head.size() + 8ull * (uint64_t)n_blocks;
  for (const auto &blk : blocks) {
    std::fwrite(&off, 8, 1, fp);
    off += blk.size();
  }
  for (const auto &blk : blocks) std::fwrite(blk.data(), 1, blk.size(), fp);
  std::fclose(fp);
}

} // namespace lrp_io
