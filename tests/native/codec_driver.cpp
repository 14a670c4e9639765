// codec_driver — host-side pieces of the CLI (codecs, JSON, lens config, rotation / lens producers)
// and the oracle, driven without a GPU.  Built twice by tests/test_sanitizers.py: plain, and with
// -fsanitize=address,undefined (sanitizers on the CPU build only).
//   codec_driver decode <file> <out.f32>            prints "width height channels layout"
//   codec_driver encode <file> <w> <h> <c> <in.f32> (format from the extension)
//   codec_driver selftest <dir>                     round trips, malformed inputs, oracle runs
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "lrp.h"
#include "lrp_config.h"
#include "lrp_image_io.h"
#include "lrp_json.h"
extern "C" {
#include "lrp_oracle.h"
}

namespace {

bool ends_with(const std::string &s, const char *suffix) {
  const size_t n = std::strlen(suffix);
  return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

lrp_io::Frame decode(const std::string &path) {
  if (ends_with(path, ".exr")) return lrp_io::read_exr(path);
  if (ends_with(path, ".png")) return lrp_io::read_png(path);
  return lrp_io::read_jpeg(path);
}

void encode(const lrp_io::Frame &f, const std::string &path) {
  if (ends_with(path, ".exr"))
    lrp_io::save_exr(f, path);
  else if (ends_with(path, ".png"))
    lrp_io::save_png(f, path);
  else
    lrp_io::save_jpeg(f, path);
}

std::vector<uint8_t> slurp(const std::string &path) {
  std::ifstream in(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}
void spill(const std::string &path, const std::vector<uint8_t> &b) {
  std::ofstream out(path, std::ios::binary);
  out.write(reinterpret_cast<const char *>(b.data()), (std::streamsize)b.size());
}

lrp_io::Frame noise(int w, int h, int c, uint32_t seed) {
  lrp_io::Frame f;
  f.width = w;
  f.height = h;
  f.channels = c;
  f.data_layout = c == 3 ? 0 : (c == 4 ? 1 : 3);
  f.data.resize((size_t)w * h * c);
  uint32_t s = seed;
  for (float &v : f.data) {
    s = s * 1664525u + 1013904223u;
    v = (float)(s >> 21) / 2048.0f;
  }
  return f;
}

int fail(const char *what) {
  std::printf("FAIL: %s\n", what);
  return 1;
}

// Every way of damaging a valid file must end in an exception (or a clean decode), never in a crash,
// a hang or an out-of-bounds access.
int damage_and_decode(const std::string &good, const std::string &scratch) {
  const std::vector<uint8_t> bytes = slurp(good);
  int rejected = 0, decoded = 0;
  auto attempt = [&](const std::vector<uint8_t> &b) {
    spill(scratch, b);
    try {
      (void)decode(scratch);
      ++decoded;
    } catch (const std::exception &) {
      ++rejected;
    }
  };
  for (size_t cut = 0; cut < bytes.size(); cut += (bytes.size() < 600 ? 1 : bytes.size() / 300)) // truncations
    attempt(std::vector<uint8_t>(bytes.begin(), bytes.begin() + (long)cut));
  uint32_t s = 12345u;
  for (int i = 0; i < 400; ++i) { // corrupted header / offset table / block sizes
    std::vector<uint8_t> b = bytes;
    s = s * 1664525u + 1013904223u;
    const size_t pos = (s >> 8) % std::min<size_t>(b.size(), 700);
    s = s * 1664525u + 1013904223u;
    b[pos] = (uint8_t)(s >> 24);
    if (i % 3 == 0 && pos + 4 <= b.size()) std::memset(&b[pos], (i & 1) ? 0xff : 0x80, 4); // huge / negative 32-bit fields
    attempt(b);
  }
  std::printf("%s: %d damaged files rejected, %d still decodable\n", good.c_str(), rejected, decoded);
  return rejected > 0 ? 0 : 1;
}

int selftest(const std::string &dir) {
  // codecs: write -> read -> write gives identical files; PNG / JPEG quantise, EXR goes through half
  for (int c : {3, 4, 5}) {
    const lrp_io::Frame f = noise(37, 23, c, 7u + (uint32_t)c);
    const std::string a = dir + "/a" + std::to_string(c) + ".exr", b = dir + "/b" + std::to_string(c) + ".exr";
    lrp_io::save_exr(f, a);
    const lrp_io::Frame g = lrp_io::read_exr(a);
    if (g.width != 37 || g.height != 23 || g.channels != c) return fail("exr geometry");
    for (size_t i = 0; i < f.data.size(); ++i)
      if (g.data[i] != lrp_io::half_to_float(lrp_io::float_to_half(f.data[i]))) return fail("exr values");
    lrp_io::save_exr(g, b);
    if (slurp(a) != slurp(b)) return fail("exr re-encode differs");
    if (damage_and_decode(a, dir + "/damaged.exr")) return fail("no damaged exr was rejected");
  }
  {
    const lrp_io::Frame f = noise(41, 19, 3, 99u);
    lrp_io::save_png(f, dir + "/a.png");
    const lrp_io::Frame g = lrp_io::read_png(dir + "/a.png");
    if (g.width != 41 || g.height != 19 || g.channels != 3) return fail("png geometry");
    lrp_io::save_png(g, dir + "/b.png");
    const lrp_io::Frame h = lrp_io::read_png(dir + "/b.png");
    if (h.data != g.data) return fail("png second generation differs"); // quantisation is idempotent
    if (damage_and_decode(dir + "/a.png", dir + "/damaged.png")) return fail("no damaged png was rejected");
  }
  try {
    const lrp_io::Frame f = noise(48, 32, 3, 5u);
    lrp_io::save_jpeg(f, dir + "/a.jpg");
    const lrp_io::Frame g = lrp_io::read_jpeg(dir + "/a.jpg");
    if (g.width != 48 || g.height != 32 || g.channels != 3) return fail("jpeg geometry");
    for (float v : g.data)
      if (!(v >= 0.0f && v <= 1.0f)) return fail("jpeg value range");
    if (damage_and_decode(dir + "/a.jpg", dir + "/damaged.jpg")) return fail("no damaged jpeg was rejected");
  } catch (const std::exception &e) {
    if (std::strstr(e.what(), "JPEG support unavailable") == nullptr) throw;
    std::printf("jpeg skipped: %s\n", e.what());
  }
  // half conversion: every half value survives the round trip, every float rounds to a neighbour
  for (uint32_t h = 0; h < 65536u; ++h) {
    const float v = lrp_io::half_to_float((uint16_t)h);
    if (v != v) continue;
    if (lrp_io::float_to_half(v) != (uint16_t)h) return fail("half round trip");
  }
  // JSON + lens config: parse, edit, dump, parse again; every lens model through both directions
  {
    const char *text = "{\"camera\": {\"type\": \"PANO\", \"panorama_type\": \"EQUIRECTANGULAR\", \"latitude_min\": -1.5, "
                       "\"latitude_max\": 1.5, \"longitude_min\": -3.0, \"longitude_max\": 3.0}, \"sensor_size\": [36.0, 24.0], "
                       "\"resolution\": [640, 480], \"frames\": [{\"name\": \"a\\u00e9\\n\"}], \"extra\": [1, 2.5e3, true, null]}";
    lrp_json::Value v = lrp_json::parse(text);
    const lrp_lens lens = lrp_cfg::extract_lens_info_from_config(v);
    if (lens.type != LRP_EQUIRECTANGULAR || lens.u.equirectangular.longitude_max != 3.0f) return fail("config read");
    for (int t : {LRP_RECTILINEAR, LRP_FISHEYE_EQUIDISTANT, LRP_FISHEYE_EQUISOLID, LRP_EQUIRECTANGULAR}) {
      lrp_lens l;
      std::memset(&l, 0, sizeof(l));
      l.type = t;
      l.sensor_width = 36.0f;
      l.sensor_height = 24.0f;
      for (int i = 0; i < 4; ++i) l.u.raw[i] = 0.25f * (float)(i + 1);
      lrp_json::Value out = v;
      lrp_cfg::store_lens_info_in_config(l, out);
      const lrp_json::Value again = lrp_json::parse(out.dump(2));
      if (t == LRP_EQUIRECTANGULAR) continue; // written as "RECTILINEAR" (sic), not readable back: reference behaviour
      const lrp_lens back = lrp_cfg::extract_lens_info_from_config(again);
      if (back.type != t || back.u.raw[0] != l.u.raw[0]) return fail("config round trip");
    }
    for (const char *bad : {"", "{", "{\"a\": }", "[1, 2", "\"\\u12\"", "{\"a\": 1} x", "nul", "-", "1e", "[\"\\x\"]"}) {
      try {
        (void)lrp_json::parse(bad);
        return fail("malformed JSON accepted");
      } catch (const std::exception &) {
      }
    }
  }
  // host-side producers + the oracle on small frames: every lens pair, sampler, odd sizes, NaN centres
  {
    float rot[9];
    lrp_rotation_matrix(0.5f, -0.25f, 0.1f, rot);
    lrp_lens lenses[3];
    const int in_w = 45, in_h = 31, out_w = 39, out_h = 27;
    for (int li = 0; li < 3; ++li)
      for (int lo = 0; lo < 3; ++lo)
        for (int interp = 0; interp < 3; ++interp)
          for (int c : {1, 3, 4, 5, 9}) {
            lrp_lens_rectilinear(&lenses[0], 18.0f, 36.0f, (float)(li == 0 ? in_w : out_w), (float)(li == 0 ? in_h : out_h));
            lrp_lens_equidistant(&lenses[1], 3.14159265f);
            lrp_lens_equirectangular_full(&lenses[2]);
            const lrp_io::Frame src = noise(in_w, in_h, c, 11u);
            std::vector<float> dst((size_t)out_w * out_h * c, -1.0f);
            lrpo_image in, out;
            std::memset(&in, 0, sizeof(in));
            std::memset(&out, 0, sizeof(out));
            std::memcpy(&in.lens, &lenses[li], sizeof(lrp_lens));
            std::memcpy(&out.lens, &lenses[lo], sizeof(lrp_lens));
            in.width = in_w;
            in.height = in_h;
            out.width = out_w;
            out.height = out_h;
            in.channels = out.channels = c;
            in.data = const_cast<float *>(src.data.data());
            out.data = dst.data();
            for (int ns = 1; ns <= 2; ++ns)
              if (lrpo_reproject(&in, &out, ns, interp, (interp & 1) ? rot : nullptr) != 0) return fail("oracle status");
            lrpo_post_process(&out, 2.0f, 4.0f);
          }
  }
  std::printf("selftest ok\n");
  return 0;
}

} // namespace

int main(int argc, char **argv) {
  try {
    const std::string mode = argc > 1 ? argv[1] : "";
    if (mode == "decode" && argc == 4) {
      const lrp_io::Frame f = decode(argv[2]);
      std::ofstream out(argv[3], std::ios::binary);
      out.write(reinterpret_cast<const char *>(f.data.data()), (std::streamsize)(f.data.size() * sizeof(float)));
      std::printf("%d %d %d %d\n", f.width, f.height, f.channels, f.data_layout);
      return 0;
    }
    if (mode == "encode" && argc == 7) {
      lrp_io::Frame f;
      f.width = std::atoi(argv[3]);
      f.height = std::atoi(argv[4]);
      f.channels = std::atoi(argv[5]);
      f.data_layout = f.channels == 3 ? 0 : (f.channels == 4 ? 1 : 3);
      const std::vector<uint8_t> raw = slurp(argv[6]);
      if (raw.size() != (size_t)f.width * f.height * f.channels * 4) return fail("input size");
      f.data.resize(raw.size() / 4);
      std::memcpy(f.data.data(), raw.data(), raw.size());
      encode(f, argv[2]);
      return 0;
    }
    if (mode == "selftest" && argc == 3) return selftest(argv[2]);
    std::printf("usage: codec_driver decode <file> <out.f32> | encode <file> <w> <h> <c> <in.f32> | selftest <dir>\n");
    return 2;
  } catch (const std::exception &e) {
    std::printf("Error: %s\n", e.what());
    return 3;
  }
}
