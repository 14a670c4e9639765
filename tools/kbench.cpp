// kbench.cpp — native kernel bench over the C ABI (include/lrp.h); no Python,
// no torch.  Times lrp_reproject_device on device-resident synthetic frames with
// HIP events on the launch stream, prints per-workload kernel time, Gpix/s and
// the algorithmic-bytes roofline fraction, plus an FNV-1a checksum of the output
// so two kernel variants (--set kernel=1|2) can be compared bit for bit at full
// size.
//
// Build (tools/build_kbench.sh):
//   hipcc -O2 -std=c++17 tools/kbench.cpp -Iinclude -L<pkg>/lib -llrp_hip -Wl,-rpath,<pkg>/lib -o tools/kbench
// Usage: kbench [--size N] [--reps R] [--warmup W] [--distinct D] [--channels C] [--ns S] [--batch B] [--geo 0|1] [--set name=value] [--sum] [--streams S] [workload ...]
//   --streams S: launch i goes to stream i % S (no events in between); the figure is wall time per launch between the first launch and the
//   last stream's completion — S = 1 and S = 2 compare back-to-back launches on one stream with launches whose tails and heads may overlap
//   --batch B: every launch renders B frames (lrp_reproject_batch_device, B <= distinct); times are per launch / B
//   --geo 0: single launches compute their coordinates in every launch (geometry cache off); --set: lrp_debug_set
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lrp.h"

#define HIP_OK(x)                                                                                  \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) {                                                                        \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
      exit(2);                                                                                     \
    }                                                                                              \
  } while (0)
#define LRP_OKAY(x)                                                                                \
  do {                                                                                             \
    int s_ = (x);                                                                                  \
    if (s_ != LRP_OK) {                                                                            \
      fprintf(stderr, "lrp error %d (%s): %s at %s:%d\n", s_, lrp_strerror(s_), lrp_last_error(), __FILE__, __LINE__); \
      exit(3);                                                                                     \
    }                                                                                              \
  } while (0)

struct Workload {
  const char *name;
  const char *in_lens, *out_lens; // "rect" | "eqd" | "eqr" | "eqrp"
  int interp;
  int has_rot;
  float rot_deg[3];
};

static const Workload kWorkloads[] = {
    {"eqd_rect_bc", "eqd", "rect", 2, 0, {0, 0, 0}},       // BASELINE configs[1] (equidistant stands in for equisolid)
    {"eqd_rect_bc_id", "eqd", "rect", 2, 1, {0, 0, 0}},    // same with the CLI's always-present identity matrix
    {"eqr_rect_bc", "eqr", "rect", 2, 1, {0, 0, 0}},       // north_star roofline case
    {"eqr_rect_bl", "eqr", "rect", 1, 1, {0, 0, 0}},
    {"eqr_rect_nn", "eqr", "rect", 0, 1, {0, 0, 0}},       // configs[0] shape
    {"eqr_eqd_bl_rot", "eqr", "eqd", 1, 1, {30, -15, 5}},  // configs[2]
    {"rect_eqr_bc", "rect", "eqr", 2, 1, {0, 0, 0}},       // configs[3] shape (without post)
    {"eqr_rect_bc_rot", "eqr", "rect", 2, 1, {90, 0, 0}},  // configs[4] side face (pan)
    {"eqr_rect_bc_pitch", "eqr", "rect", 2, 1, {0, 90, 0}}, // configs[4] top face (pitch: looks at the pole)
    {"rect_rect_bc", "rect", "rect", 2, 1, {10, 5, 0}},
    {"eqd_eqd_bc", "eqd", "eqd", 2, 1, {10, 5, 0}},
    {"eqr_eqr_bc_rot", "eqr", "eqr", 2, 1, {30, -15, 5}},
    {"eqr_eqd_bc_rot", "eqr", "eqd", 2, 1, {30, -15, 5}},  // SURVEY 8d scaling shape: configs[2] with bicubic
    {"eqr_rect_bc_gen", "eqr", "rect", 2, 1, {30, -15, 5}}, // general rotation (plain blocks)
    {"rect_eqd_bc", "rect", "eqd", 2, 0, {0, 0, 0}},
    {"eqd_rect_bl", "eqd", "rect", 1, 0, {0, 0, 0}},
    {"eqd_rect_nn", "eqd", "rect", 0, 0, {0, 0, 0}},
    {"rect_eqr_nn", "rect", "eqr", 0, 1, {0, 0, 0}},
    {"rect_eqr_bl", "rect", "eqr", 1, 1, {0, 0, 0}},
    // partial panorama as the target (KBENCH_EQRP=lon_min,lon_max,lat_min,lat_max picks the window)
    {"rect_eqrp_nn", "rect", "eqrp", 0, 1, {0, 0, 0}},
    {"rect_eqrp_bl", "rect", "eqrp", 1, 1, {0, 0, 0}},
    {"rect_eqrp_bc", "rect", "eqrp", 2, 1, {0, 0, 0}},
    // general rotation, nearest / bilinear: the plain (unmirrored) path of the tile kernels
    {"eqr_rect_nn_rot", "eqr", "rect", 0, 1, {30, -15, 5}},
    {"eqr_rect_bl_rot", "eqr", "rect", 1, 1, {30, -15, 5}},
};

static void make_lens(lrp_lens *L, const char *kind, int w, int h) {
  if (!strcmp(kind, "rect"))
    lrp_lens_rectilinear(L, 18.0f, 36.0f, (float)w, (float)h);
  else if (!strcmp(kind, "eqd"))
    lrp_lens_equidistant(L, 3.14159265f);
  else if (!strcmp(kind, "eqrp")) {
    float v[4] = {-1.0f, 1.5f, -0.6f, 0.7f};
    if (const char *e = getenv("KBENCH_EQRP")) sscanf(e, "%f,%f,%f,%f", &v[0], &v[1], &v[2], &v[3]);
    lrp_lens_equirectangular(L, v[0], v[1], v[2], v[3]);
  }
  else
    lrp_lens_equirectangular_full(L);
}

static uint64_t fnv1a(const void *p, size_t n) {
  const uint64_t *q = (const uint64_t *)p;
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n / 8; ++i) {
    h ^= q[i];
    h *= 1099511628211ull;
  }
  return h;
}

int main(int argc, char **argv) {
  int size = 4096, reps = 20, distinct = 4, channels = 4, ns = 1, out_size = 0, warmup = 100, batch = 0, n_streams = 0;
  bool sum = false, post = false;
  std::vector<std::string> names;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() { return (i + 1 < argc) ? atoi(argv[++i]) : 0; };
    if (a == "--size") size = next();
    else if (a == "--out-size") out_size = next();
    else if (a == "--reps") reps = next();
    else if (a == "--warmup") warmup = next();
    else if (a == "--distinct") distinct = next();
    else if (a == "--channels") channels = next();
    else if (a == "--ns") ns = next();
    else if (a == "--batch") batch = next();
    else if (a == "--streams") n_streams = next();
    else if (a == "--geo") lrp_debug_set("geo_cache", next());
    else if (a == "--set" && i + 1 < argc) {
      std::string kv = argv[++i];
      const size_t eq = kv.find('=');
      if (eq == std::string::npos || lrp_debug_set(kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1)) < 0) {
        fprintf(stderr, "bad --set %s\n", kv.c_str());
        return 1;
      }
    }
    else if (a == "--sum") sum = true;
    else if (a == "--post") post = true;
    else names.push_back(a);
  }
  if (!out_size) out_size = size;
  if (batch > 0 && distinct < batch) distinct = batch;
  if (names.empty())
    for (const auto &w : kWorkloads) names.push_back(w.name);
  if (lrp_device_count() < 1) {
    fprintf(stderr, "no HIP device\n");
    return 1;
  }
  HIP_OK(hipSetDevice(0));
  hipStream_t stream;
  HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  const size_t in_elems = (size_t)size * size * channels, out_elems = (size_t)out_size * out_size * channels;
  std::vector<float *> src(distinct), dst(distinct);
  // KBENCH_ONE_ALLOC=1: all frames are slices of two allocations (an address-translation experiment: one large
  // allocation can be mapped with larger page fragments than many 256 MiB ones)
  const bool one_alloc = getenv("KBENCH_ONE_ALLOC") && atoi(getenv("KBENCH_ONE_ALLOC")) != 0;
  float *src_all = nullptr, *dst_all = nullptr;
  if (one_alloc) {
    HIP_OK(hipMalloc(&src_all, in_elems * 4 * (size_t)distinct));
    HIP_OK(hipMalloc(&dst_all, out_elems * 4 * (size_t)distinct));
  }
  for (int i = 0; i < distinct; ++i) {
    if (one_alloc) {
      src[i] = src_all + in_elems * (size_t)i;
      dst[i] = dst_all + out_elems * (size_t)i;
    } else {
      HIP_OK(hipMalloc(&src[i], in_elems * 4));
      HIP_OK(hipMalloc(&dst[i], out_elems * 4));
    }
    LRP_OKAY(lrp_synth_fill_device(src[i], size, size, channels, 0x5EED0000u + i, channels == 5 ? 4 : -1, 0, stream));
  }
  HIP_OK(hipStreamSynchronize(stream));
  std::vector<float> host;
  if (sum) host.resize(out_elems);
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  printf("# size %d -> %d, C=%d, ns=%d, reps=%d, distinct=%d, batch=%d, kernel family %d, geo_cache=%d\n", size, out_size, channels, ns, reps,
         distinct, batch, lrp_debug_set("kernel", -1), lrp_debug_set("geo_cache", -1));
  for (const auto &nm : names) {
    const Workload *W = nullptr;
    for (const auto &w : kWorkloads)
      if (nm == w.name) W = &w;
    if (!W) {
      fprintf(stderr, "unknown workload %s\n", nm.c_str());
      continue;
    }
    lrp_image in{}, out{};
    make_lens(&in.lens, W->in_lens, size, size);
    make_lens(&out.lens, W->out_lens, out_size, out_size);
    in.width = in.height = size;
    out.width = out.height = out_size;
    in.channels = out.channels = channels;
    float rot[9];
    const float d2r = 3.14159265358979f / 180.0f;
    lrp_rotation_matrix(W->rot_deg[0] * d2r, W->rot_deg[1] * d2r, W->rot_deg[2] * d2r, rot);
    lrp_post pp{2.0f, 4.0f};
    std::vector<lrp_image> ins((size_t)(batch > 0 ? batch : 0), in), outs((size_t)(batch > 0 ? batch : 0), out);
    for (int b = 0; b < batch; ++b) {
      // KBENCH_BATCH_DISTINCT=D: the batch cycles over D frame pairs (a footprint experiment; frames rendered more than once)
      // (KBENCH_BATCH_DISTINCT_SRC / _DST: the same for the sources / the destinations alone)
      auto cycle = [&](const char *name) {
        const char *v = getenv(name) ? getenv(name) : getenv("KBENCH_BATCH_DISTINCT");
        const int c = v ? atoi(v) : batch;
        return c > 0 ? c : batch;
      };
      ins[(size_t)b].data = src[(size_t)(b % cycle("KBENCH_BATCH_DISTINCT_SRC"))];
      outs[(size_t)b].data = dst[(size_t)(b % cycle("KBENCH_BATCH_DISTINCT_DST"))];
    }
    auto launch = [&](int i) {
      if (batch > 0) {
        LRP_OKAY(lrp_reproject_batch_device(ins.data(), outs.data(), batch, ns, W->interp, W->has_rot ? rot : nullptr,
                                            post ? &pp : nullptr, 0, stream));
        return;
      }
      in.data = src[i % distinct];
      out.data = dst[i % distinct];
      LRP_OKAY(lrp_reproject_device(&in, &out, ns, W->interp, W->has_rot ? rot : nullptr, post ? &pp : nullptr, 0, stream));
    };
    if (n_streams > 0) { // wall time per launch over S streams
      std::vector<hipStream_t> ss((size_t)n_streams);
      for (auto &q : ss) HIP_OK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
      auto launch_on = [&](int i) {
        in.data = src[i % distinct];
        out.data = dst[i % distinct];
        LRP_OKAY(lrp_reproject_device(&in, &out, ns, W->interp, W->has_rot ? rot : nullptr, post ? &pp : nullptr, 0, ss[(size_t)(i % n_streams)]));
      };
      for (int i = 0; i < warmup; ++i) launch_on(i);
      HIP_OK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        HIP_OK(hipEventRecord(e0, ss[0]));
        for (int k = 1; k < n_streams; ++k) HIP_OK(hipStreamWaitEvent(ss[(size_t)k], e0, 0));
        for (int i = 0; i < reps; ++i) launch_on(i);
        for (int k = 1; k < n_streams; ++k) { // join into stream 0
          hipEvent_t j;
          HIP_OK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
          HIP_OK(hipEventRecord(j, ss[(size_t)k]));
          HIP_OK(hipStreamWaitEvent(ss[0], j, 0));
          HIP_OK(hipEventDestroy(j));
        }
        HIP_OK(hipEventRecord(e1, ss[0]));
        HIP_OK(hipEventSynchronize(e1));
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("%-18s %d streams: %8.1f us per launch (best of 5 runs of %d launches)\n", W->name, n_streams, best * 1e3 / reps, reps);
      for (auto &q : ss) HIP_OK(hipStreamDestroy(q));
      continue;
    }
    if (batch > 0 && warmup > 100 / batch) warmup = 100 / batch + 2;
    // ~20 ms of back-to-back launches first: the chip settles its clock over ~10 ms of load
    for (int i = 0; i < warmup; ++i) launch(i);
    HIP_OK(hipStreamSynchronize(stream));
    // all launches queued back to back, one event pair each (no host sync in between:
    // the chip stays at its loaded clock, as in a batch run)
    std::vector<hipEvent_t> ev((size_t)reps + 1);
    for (auto &e : ev) HIP_OK(hipEventCreate(&e));
    HIP_OK(hipEventRecord(ev[0], stream));
    for (int i = 0; i < reps; ++i) {
      launch(i);
      HIP_OK(hipEventRecord(ev[(size_t)i + 1], stream));
    }
    HIP_OK(hipStreamSynchronize(stream));
    float best = 1e30f, total = 0;
    for (int i = 0; i < reps; ++i) {
      float ms;
      HIP_OK(hipEventElapsedTime(&ms, ev[(size_t)i], ev[(size_t)i + 1]));
      total += ms;
      if (ms < best) best = ms;
    }
    if (getenv("KBENCH_SERIES")) {
      printf("    series(us):");
      for (int i = 0; i < reps; ++i) {
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, ev[(size_t)i], ev[(size_t)i + 1]));
        printf(" %.0f", ms * 1e3);
      }
      printf("\n");
    }
    for (auto &e : ev) HIP_OK(hipEventDestroy(e));
    const double avg_s = total / reps * 1e-3 / (batch > 0 ? batch : 1); // per frame
    best /= (float)(batch > 0 ? batch : 1);
    const double bytes = (double)(in_elems + out_elems) * 4;
    uint64_t h = 0;
    if (sum) {
      launch(0);
      HIP_OK(hipStreamSynchronize(stream));
      HIP_OK(hipMemcpy(host.data(), dst[0], out_elems * 4, hipMemcpyDeviceToHost));
      h = fnv1a(host.data(), out_elems * 4);
    }
    if (auto rd = (void (*)(unsigned *))dlsym(RTLD_DEFAULT, "lrp_debug_read_tiers")) { // diagnostic builds only
      unsigned t[4] = {0, 0, 0, 0};
      rd(t);
      printf("    window-kernel blocks per tier (all launches so far): coefficients %u, raw taps %u, direct %u\n", t[0], t[1], t[2]);
    }
    printf("%-18s avg %8.1f us  min %8.1f us  %8.2f Gpix/s  %7.1f GB/s algorithmic  frac %.3f", W->name, avg_s * 1e6,
           best * 1e3, (double)out_size * out_size / avg_s / 1e9, bytes / avg_s / 1e9, bytes / avg_s / 8e12);
    if (sum) printf("  fnv %016llx", (unsigned long long)h);
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
