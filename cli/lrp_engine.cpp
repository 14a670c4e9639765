// lrp_engine.cpp — see lrp_engine.h.  The per-file sequence is the reference worker's
// (src/main.cpp:541-620): skip-if-exists, read, reproject (+ post_process), write, progress line.
#include "lrp_engine.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "lrp_half.h"
#include "lrp_image_io.h"

namespace fs = std::filesystem;

namespace lrp_cli {

namespace {

struct PinnedBuffer { // a grow-only page-locked buffer, reused by a worker from file to file
  void *ptr = nullptr;
  size_t cap = 0;
  bool lent = false;
  ~PinnedBuffer() { lrp_host_free(ptr); }
  void *reserve(size_t n) {
    if (n > cap) {
      lrp_host_free(ptr);
      ptr = nullptr;
      cap = 0;
      if (lrp_host_alloc(&ptr, n) != LRP_OK) throw std::runtime_error(std::string("page-locked allocation failed: ") + lrp_last_error());
      cap = n;
    }
    return ptr;
  }
};

// Decoded input files land in the worker's own grow-only page-locked buffer: no hipHostMalloc / hipHostFree per file
// (page-locking tens of MB costs milliseconds, and a free may wait for the device's streams — i.e. for the other
// workers' uploads, kernels and downloads on the shared pipeline).  A worker holds one decoded file at a time; should
// a second allocation arrive while the buffer is lent out it gets page-locked memory of its own.
thread_local PinnedBuffer t_input;
void *pinned_alloc(size_t n) {
  if (!t_input.lent) {
    try {
      void *p = t_input.reserve(n);
      t_input.lent = true;
      return p;
    } catch (const std::exception &) {
      return nullptr;
    }
  }
  void *p = nullptr;
  return lrp_host_alloc(&p, n) == LRP_OK ? p : nullptr;
}
void pinned_free(void *p) {
  if (p != nullptr && p == t_input.ptr)
    t_input.lent = false; // stays page-locked for the next file
  else
    lrp_host_free(p);
}
const lrp_io::Allocator kPinned{pinned_alloc, pinned_free};

struct Aborted {}; // thrown by check_status once the abort message has been printed; nothing else uses this type

struct Shared {
  const RunPlan &plan;
  int total;
  lrp_io::Allocator input_memory; // page-locked when a GPU takes part, the heap for a pure copy run
  std::atomic_int done{0}, failed{0};
  std::atomic_bool abort{false};
  std::mutex abort_mutex; // serialises the one abort message
};

struct Device {
  int index = 0;
  lrp_context *pipeline = nullptr; // shared by the workers of this device (submissions are serialised inside)
  std::atomic_size_t next{0};      // next file of this device's block
  size_t begin = 0, end = 0;
};

lrp_image describe(const lrp_lens &lens, int w, int h, int c, int layout, void *data) {
  lrp_image im;
  std::memset(&im, 0, sizeof(im));
  im.lens = lens;
  im.width = w;
  im.height = h;
  im.channels = c;
  im.data = static_cast<float *>(data);
  im.data_layout = layout;
  return im;
}

// unsupported lens / interpolation: the reference prints the message and ends the process with status 1
// (src/reproject.cpp:365-366,396-397,416-417); here the message is printed once, the workers stop taking
// files and main() returns 1 after they have joined (no exit() under running threads)
void check_status(Shared &sh, int st) {
  if (st == LRP_OK) return;
  if (st == LRP_ERR_OUTPUT_LENS || st == LRP_ERR_INPUT_LENS || st == LRP_ERR_INTERPOLATION) {
    std::lock_guard<std::mutex> lock(sh.abort_mutex);
    if (!sh.abort.exchange(true)) std::printf("%s\n", lrp_strerror(st));
    throw Aborted{};
  }
  throw std::runtime_error(std::string(lrp_strerror(st)) + ": " + lrp_last_error());
}

// The copy path (--no-reproject at scale 1, src/main.cpp:592-595) and the outputs no packed format
// carries (PNG and EXR of one image, PNG of a five-channel image) work on float frames on the host.
void process_on_host_floats(Shared &sh, Device &dev, const lrp_io::Packed &input, const fs::path &png, const fs::path &exr) {
  const RunPlan &plan = sh.plan;
  lrp_io::Frame in = lrp_io::unpack(input), out;
  out.width = plan.out_width;
  out.height = plan.out_height;
  out.channels = in.channels;
  out.data_layout = in.data_layout;
  out.data.resize((size_t)out.width * out.height * out.channels);
  lrp_image cin = describe(plan.input_lens, in.width, in.height, in.channels, in.data_layout, in.data.data());
  lrp_image cout = describe(plan.output_lens, out.width, out.height, out.channels, out.data_layout, out.data.data());
  const lrp_post pp{(float)plan.exposure, (float)plan.reinhard}; // narrowed at the call, src/main.cpp:602
  if (plan.copies_pixels()) {
    if (in.data.size() < out.data.size()) throw std::runtime_error("input smaller than the configured resolution");
    std::memcpy(out.data.data(), in.data.data(), out.data.size() * sizeof(float));
    if (plan.post_process()) check_status(sh, lrp_post_process(&cout, pp.exposure, pp.reinhard, dev.index));
  } else {
    check_status(sh, lrp_reproject(&cin, &cout, plan.num_samples, plan.interpolation, plan.rotation,
                                   plan.post_process() ? &pp : nullptr, dev.index));
  }
  if (plan.write_png) lrp_io::save_png(out, png.string());
  if (plan.write_exr) lrp_io::save_exr(out, exr.string());
}

void process_file(Shared &sh, Device &dev, PinnedBuffer &out_buffer, const fs::path &path) {
  const RunPlan &plan = sh.plan;
  fs::path png = plan.output_dir / path.filename(), exr = png;
  png.replace_extension(".png");
  exr.replace_extension(".exr");
  if (plan.skip_if_exists && (!plan.write_png || fs::exists(png)) && (!plan.write_exr || fs::exists(exr))) {
    std::printf("Skipping '%s'. Already exists.\n", png.c_str());
    ++sh.done;
    return;
  }
  const std::string ext = path.extension().string();
  if (ext != ".exr" && ext != ".png" && ext != ".jpeg" && ext != ".jpg") {
    std::printf("Input format not supported: %s\n", ext.c_str());
    return; // the reference carries on with an uninitialised image here
  }
  const lrp_io::Packed input = lrp_io::read_packed(path.string(), sh.input_memory);
  // the run's output size, not the file's (src/main.cpp:581-587)
  if (plan.out_width < 1 || plan.out_height < 1) throw std::runtime_error("empty output image");
  const int channels = input.channels;
  const bool one_format = plan.write_png != plan.write_exr;
  const bool png_packable = channels <= 4; // a fifth channel would be quantised into the alpha byte save_png forces to 255
  if (plan.copies_pixels() || !one_format || (plan.write_png && !png_packable)) {
    process_on_host_floats(sh, dev, input, png, exr);
  } else {
    // the frame goes up in its file format and comes back in the output file's
    const int out_format = plan.write_png ? LRP_PIXEL_U8_GAMMA : LRP_PIXEL_F16;
    const int out_packed = plan.write_png ? 4 : channels;
    const size_t out_bytes = (size_t)plan.out_width * plan.out_height * out_packed * (plan.write_png ? 1 : 2);
    void *out_pixels = out_buffer.reserve(out_bytes);
    lrp_image cin = describe(plan.input_lens, input.width, input.height, channels, input.data_layout, input.bytes);
    lrp_image cout = describe(plan.output_lens, plan.out_width, plan.out_height, channels, input.data_layout, out_pixels);
    const lrp_post pp{(float)plan.exposure, (float)plan.reinhard};
    int ticket = -1;
    check_status(sh, lrp_context_submit_packed(dev.pipeline, &cin, input.format, input.packed_channels, &cout, out_format, out_packed,
                                               255u, plan.num_samples, plan.interpolation, plan.rotation,
                                               plan.post_process() ? &pp : nullptr, &ticket));
    check_status(sh, lrp_context_wait_ticket(dev.pipeline, ticket));
    if (plan.write_png)
      lrp_io::save_png_rgba8(static_cast<const uint8_t *>(out_pixels), plan.out_width, plan.out_height, png.string());
    else
      lrp_io::save_exr_half(static_cast<const uint16_t *>(out_pixels), plan.out_width, plan.out_height, channels, exr.string());
  }
  const int n = ++sh.done;
  std::printf("%4d / %4d: %s\n", n, sh.total, path.stem().c_str());
}

} // namespace

RunResult run_files(const RunPlan &plan, const std::vector<fs::path> &files) {
  RunResult result;
  // A pure copy run touches no pixel arithmetic and needs no GPU; everything else does.
  const bool needs_gpu = !plan.copies_pixels() || plan.post_process();
  const int n_dev = lrp_device_count();
  if (needs_gpu && n_dev < 1) {
    std::printf("Error: no usable HIP device (this build has no CPU path).\n");
    result.aborted = true;
    return result;
  }
  if (needs_gpu && (plan.first_device < 0 || plan.first_device >= n_dev)) {
    std::printf("Error: --device %d is out of range (%d HIP device%s visible).\n", plan.first_device, n_dev, n_dev == 1 ? "" : "s");
    result.aborted = true;
    return result;
  }
  const int gpus = needs_gpu ? std::max(1, std::min({plan.gpus, 64, n_dev - plan.first_device})) : 1;
  const int workers = std::max(plan.workers, gpus);
  // static contiguous blocks of the sorted list, one per GPU
  std::vector<Device> devices((size_t)gpus);
  const size_t per = (files.size() + (size_t)gpus - 1) / (size_t)gpus;
  for (int g = 0; g < gpus; ++g) {
    devices[(size_t)g].index = plan.first_device + g;
    devices[(size_t)g].begin = std::min(files.size(), (size_t)g * per);
    devices[(size_t)g].end = std::min(files.size(), devices[(size_t)g].begin + per);
    if (needs_gpu) {
      const int on_this_device = (workers - g + gpus - 1) / gpus; // workers t with t % gpus == g
      // one slot more than images in flight: a slot is only re-used after its owner has collected its ticket
      // (--streams asks for more — never fewer: a worker that found no free slot would wait for its own ticket)
      const int slots = std::max({3, on_this_device + 1, plan.streams});
      if (lrp_context_create(&devices[(size_t)g].pipeline, devices[(size_t)g].index, std::min(64, slots)) != LRP_OK) {
        std::printf("Error: %s: %s\n", "cannot create the GPU pipeline", lrp_last_error());
        for (Device &d : devices) lrp_context_destroy(d.pipeline);
        result.aborted = true;
        return result;
      }
    }
  }
  Shared shared{plan, (int)files.size(), needs_gpu ? kPinned : lrp_io::heap_allocator()};
  std::vector<std::thread> pool;
  for (int t = 0; t < workers; ++t) {
    pool.emplace_back([&, t] {
      Device &dev = devices[(size_t)(t % gpus)];
      PinnedBuffer out_buffer;
      while (!shared.abort.load()) {
        const size_t i = dev.begin + dev.next.fetch_add(1);
        if (i >= dev.end) break;
        try {
          process_file(shared, dev, out_buffer, files[i]);
        } catch (const Aborted &) { // message already printed; every other exception type reaches the handler below
          break;
        } catch (const std::exception &e) { // the reference worker's catch (src/main.cpp:617-619)
          std::printf("Error: %s\n", e.what());
          ++shared.failed;
        }
      }
    });
  }
  for (std::thread &th : pool) th.join();
  for (Device &d : devices) lrp_context_destroy(d.pipeline);
  result.failed = shared.failed.load();
  result.aborted = shared.abort.load();
  return result;
}

} // namespace lrp_cli
