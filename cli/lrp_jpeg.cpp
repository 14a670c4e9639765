// lrp_jpeg.cpp — JPEG in / out with the in-memory conventions of reference
// src/image_formats.cpp:26-142 (read_jpeg / save_jpeg):
//   read : channels = the decoder's output components, v = pow(p / 255, 2.2) per component   (:26-77)
//   write: uint8(255.9 * pow(clamp(v, 0, 1), 1 / 2.2)), components = channels, JCS_RGB,
//          quality 95 limited to baseline                                                     (:79-142)
// The reference reads three components per pixel whatever the file has (a greyscale JPEG makes
// it read and write out of bounds); here anything but a three-component image is an error.
//
// libjpeg itself: the image ships IJG libjpeg 9 (header /opt/conda/include/jpeglib.h, runtime
// /opt/conda/lib/libjpeg.so.9) outside the default linker / loader paths, next to a conda
// libstdc++ that must not get in front of the system one.  So the library is opened at run time
// by path — LRP_LIBJPEG, /opt/conda/lib/libjpeg.so.9, then the loader's own search for
// libjpeg.so.9 — and only its C entry points are used; libjpeg checks the structure sizes and
// version this file was compiled with against its own and the mismatch (a v8 / turbo runtime)
// arrives here as an ordinary error.
#include <cstddef>
#include <cstdio> // jpeglib.h needs size_t and FILE declared first

#include <jpeglib.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <csetjmp>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "lrp_image_io.h"

namespace lrp_io {

namespace {

struct JpegApi {
  void *handle = nullptr;
  std::string error;
  struct jpeg_error_mgr *(*std_error)(struct jpeg_error_mgr *) = nullptr;
  void (*create_decompress)(j_decompress_ptr, int, size_t) = nullptr;
  void (*stdio_src)(j_decompress_ptr, FILE *) = nullptr;
  int (*read_header)(j_decompress_ptr, boolean) = nullptr;
  boolean (*start_decompress)(j_decompress_ptr) = nullptr;
  JDIMENSION (*read_scanlines)(j_decompress_ptr, JSAMPARRAY, JDIMENSION) = nullptr;
  boolean (*finish_decompress)(j_decompress_ptr) = nullptr;
  void (*destroy_decompress)(j_decompress_ptr) = nullptr;
  void (*create_compress)(j_compress_ptr, int, size_t) = nullptr;
  void (*stdio_dest)(j_compress_ptr, FILE *) = nullptr;
  void (*set_defaults)(j_compress_ptr) = nullptr;
  void (*set_quality)(j_compress_ptr, int, boolean) = nullptr;
  void (*start_compress)(j_compress_ptr, boolean) = nullptr;
  JDIMENSION (*write_scanlines)(j_compress_ptr, JSAMPARRAY, JDIMENSION) = nullptr;
  void (*finish_compress)(j_compress_ptr) = nullptr;
  void (*destroy_compress)(j_compress_ptr) = nullptr;
};

const JpegApi &api() {
  static JpegApi a;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<std::string> names;
    if (const char *e = std::getenv("LRP_LIBJPEG")) names.push_back(e);
    names.push_back("/opt/conda/lib/libjpeg.so.9");
    names.push_back("libjpeg.so.9");
    for (const std::string &n : names) {
      a.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (a.handle) break;
    }
    if (!a.handle) {
      a.error = "no libjpeg.so.9 found (set LRP_LIBJPEG to an IJG libjpeg 9 runtime)";
      return;
    }
    bool ok = true;
    auto sym = [&](auto &fn, const char *name) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(a.handle, name));
      ok = ok && fn != nullptr;
    };
    sym(a.std_error, "jpeg_std_error");
    sym(a.create_decompress, "jpeg_CreateDecompress");
    sym(a.stdio_src, "jpeg_stdio_src");
    sym(a.read_header, "jpeg_read_header");
    sym(a.start_decompress, "jpeg_start_decompress");
    sym(a.read_scanlines, "jpeg_read_scanlines");
    sym(a.finish_decompress, "jpeg_finish_decompress");
    sym(a.destroy_decompress, "jpeg_destroy_decompress");
    sym(a.create_compress, "jpeg_CreateCompress");
    sym(a.stdio_dest, "jpeg_stdio_dest");
    sym(a.set_defaults, "jpeg_set_defaults");
    sym(a.set_quality, "jpeg_set_quality");
    sym(a.start_compress, "jpeg_start_compress");
    sym(a.write_scanlines, "jpeg_write_scanlines");
    sym(a.finish_compress, "jpeg_finish_compress");
    sym(a.destroy_compress, "jpeg_destroy_compress");
    if (!ok) a.error = "the libjpeg runtime lacks an entry point";
  });
  if (!a.error.empty()) throw std::runtime_error("JPEG support unavailable: " + a.error);
  return a;
}

// libjpeg's default error_exit calls exit(); this one returns to the caller's setjmp with the message
struct ErrorTrap {
  struct jpeg_error_mgr mgr;
  std::jmp_buf jump;
  char message[JMSG_LENGTH_MAX];
};
void trap_error(j_common_ptr cinfo) {
  ErrorTrap *t = reinterpret_cast<ErrorTrap *>(cinfo->err);
  (*cinfo->err->format_message)(cinfo, t->message);
  std::longjmp(t->jump, 1);
}

} // namespace

// libjpeg reports errors by longjmp: the frames that call into it hold plain data only (no object
// with a destructor is created between setjmp and the calls), buffers are owned by the callers below.
namespace {

// 0: ok, 1: libjpeg error (trap.message), 2: bad header, 3: not three components
int jpeg_open(const JpegApi &J, jpeg_decompress_struct *cinfo, ErrorTrap *trap, FILE *fp) {
  if (setjmp(trap->jump)) return 1;
  J.create_decompress(cinfo, JPEG_LIB_VERSION, sizeof(*cinfo));
  J.stdio_src(cinfo, fp);
  if (J.read_header(cinfo, TRUE) != JPEG_HEADER_OK) return 2;
  J.start_decompress(cinfo);
  return cinfo->output_components == 3 ? 0 : 3;
}

int jpeg_decode_rows(const JpegApi &J, jpeg_decompress_struct *cinfo, ErrorTrap *trap, JSAMPLE *out) {
  if (setjmp(trap->jump)) return 1;
  const size_t n = (size_t)cinfo->output_width * 3;
  while (cinfo->output_scanline < cinfo->output_height) {
    JSAMPROW rows[1] = {out + (size_t)cinfo->output_scanline * n};
    if (J.read_scanlines(cinfo, rows, 1) != 1) return 2;
  }
  J.finish_decompress(cinfo);
  return 0;
}

int jpeg_encode(const JpegApi &J, jpeg_compress_struct *cinfo, ErrorTrap *trap, FILE *fp, int width, int height,
                const float *data, JSAMPLE *row) {
  if (setjmp(trap->jump)) return 1;
  J.create_compress(cinfo, JPEG_LIB_VERSION, sizeof(*cinfo));
  J.stdio_dest(cinfo, fp);
  cinfo->image_width = (JDIMENSION)width;
  cinfo->image_height = (JDIMENSION)height;
  cinfo->input_components = 3;
  cinfo->in_color_space = JCS_RGB;
  J.set_defaults(cinfo);
  J.set_quality(cinfo, 95, TRUE);
  J.start_compress(cinfo, TRUE);
  const size_t n = (size_t)width * 3;
  while (cinfo->next_scanline < cinfo->image_height) {
    const float *in = data + (size_t)cinfo->next_scanline * n;
    for (size_t i = 0; i < n; ++i) {
      float s = std::max(0.0f, std::min(1.0f, in[i])); // :125-128
      s = std::pow(s, 1.0f / 2.2f);
      row[i] = (JSAMPLE)(uint8_t)(255.9f * s);
    }
    JSAMPROW rows[1] = {row};
    J.write_scanlines(cinfo, rows, 1);
  }
  J.finish_compress(cinfo);
  return 0;
}

} // namespace

Packed read_jpeg_packed(const std::string &path, const Allocator &alloc) {
  const JpegApi &J = api();
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) throw std::runtime_error("cannot open " + path);
  jpeg_decompress_struct cinfo;
  ErrorTrap trap;
  cinfo.err = J.std_error(&trap.mgr);
  trap.mgr.error_exit = trap_error;
  trap.message[0] = 0;
  auto fail = [&](const std::string &what) {
    J.destroy_decompress(&cinfo); // safe on a structure that jpeg_CreateDecompress has at least zeroed
    std::fclose(fp);
    return std::runtime_error(path + ": " + what);
  };
  cinfo.mem = nullptr; // jpeg_destroy checks it: nothing to free if creation itself failed
  int rc = jpeg_open(J, &cinfo, &trap, fp);
  if (rc == 1) throw fail(trap.message);
  if (rc == 2) throw fail("cannot read the JPEG header");
  if (rc == 3) throw fail(std::to_string(cinfo.output_components) + "-component JPEG (three components expected)");
  Packed p;
  p.width = (int)cinfo.output_width;
  p.height = (int)cinfo.output_height;
  p.channels = p.packed_channels = 3;
  p.data_layout = 0;
  p.format = 2;
  try {
    p.allocate(alloc, (size_t)p.width * p.height * 3);
  } catch (...) {
    J.destroy_decompress(&cinfo);
    std::fclose(fp);
    throw;
  }
  static_assert(sizeof(JSAMPLE) == 1, "8-bit libjpeg");
  rc = jpeg_decode_rows(J, &cinfo, &trap, p.bytes);
  if (rc == 1) throw fail(trap.message);
  if (rc == 2) throw fail("cannot read a JPEG scanline");
  J.destroy_decompress(&cinfo);
  std::fclose(fp);
  return p;
}

// v = pow(float(p) / 255.0f, 2.2f) per component (src/image_formats.cpp:64-66): unpack()'s table
Frame read_jpeg(const std::string &path) { return unpack(read_jpeg_packed(path, heap_allocator())); }

void save_jpeg(const Frame &f, const std::string &path) {
  const JpegApi &J = api();
  if (f.channels != 3) throw std::runtime_error("JPEG output needs three channels"); // JCS_RGB with input_components = channels
  FILE *fp = std::fopen(path.c_str(), "wb");
  if (!fp) throw std::runtime_error("cannot write " + path);
  jpeg_compress_struct cinfo;
  ErrorTrap trap;
  cinfo.err = J.std_error(&trap.mgr);
  trap.mgr.error_exit = trap_error;
  trap.message[0] = 0;
  cinfo.mem = nullptr;
  std::vector<JSAMPLE> row((size_t)f.width * 3);
  const int rc = jpeg_encode(J, &cinfo, &trap, fp, f.width, f.height, f.data.data(), row.data());
  J.destroy_compress(&cinfo);
  std::fclose(fp);
  if (rc != 0) throw std::runtime_error(path + ": " + trap.message);
}

} // namespace lrp_io
