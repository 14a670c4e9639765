// staged_bench.cpp — host-buffer (PCIe-inclusive) throughput of the batch path
// lrp_context_* : pinned or pageable 4096^2 RGBA frames in, frames out.
// Build: hipcc -O2 -std=c++17 tools/staged_bench.cpp -Iinclude -L<lib> -llrp_hip -o tools/staged_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include "lrp.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define LK(x) do { int s_ = (x); if (s_ != LRP_OK) { printf("lrp error %d %s (%s) line %d\n", s_, lrp_strerror(s_), lrp_last_error(), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
  const int size = 4096, c = 4, n = argc > 1 ? atoi(argv[1]) : 12;
  const size_t bytes = (size_t)size * size * c * 4;
  lrp_image in{}, out{};
  lrp_lens_equidistant(&in.lens, 3.14159265f);
  lrp_lens_rectilinear(&out.lens, 18.0f, 36.0f, size, size);
  in.width = in.height = out.width = out.height = size;
  in.channels = out.channels = c;
  for (int pinned = 0; pinned < 2; ++pinned) {
    std::vector<float *> src(n), dst(n);
    for (int i = 0; i < n; ++i) {
      if (pinned) { CK(hipHostMalloc((void **)&src[i], bytes, hipHostMallocDefault)); CK(hipHostMalloc((void **)&dst[i], bytes, hipHostMallocDefault)); }
      else { src[i] = (float *)malloc(bytes); dst[i] = (float *)malloc(bytes); }
      for (size_t k = 0; k < bytes / 4; k += 1024) src[i][k] = (float)(k & 1023) / 1024.0f; // touch pages
      memset(dst[i], 0, bytes);
    }
    for (int streams : {1, 2, 3, 4}) {
      lrp_context *ctx = nullptr;
      LK(lrp_context_create(&ctx, 0, streams));
      double best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {
          in.data = src[i]; out.data = dst[i];
          LK(lrp_context_submit(ctx, &in, &out, 1, LRP_BICUBIC, nullptr, nullptr));
        }
        LK(lrp_context_wait(ctx));
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt < best) best = dt;
      }
      printf("pinned=%d streams=%d: %.0f Mpix/s staged, %.2f ms/frame, %.1f GB/s PCIe (both directions)\n", pinned, streams,
             (double)n * size * size / best / 1e6, best / n * 1e3, 2.0 * n * bytes / best / 1e9);
      lrp_context_destroy(ctx);
    }
    for (int i = 0; i < n; ++i) { if (pinned) { hipHostFree(src[i]); hipHostFree(dst[i]); } else { free(src[i]); free(dst[i]); } }
  }
  // The same frames in their FILE formats (lrp_context_submit_packed, page-locked buffers): the decode /
  // encode kernels run on the device, PCIe carries 8 (binary16 RGBA) or 4 (RGBA8) bytes per pixel each way.
  struct { const char *name; int fmt; size_t px_bytes; } formats[] = {{"binary16 RGBA (EXR)", LRP_PIXEL_F16, 8}, {"RGBA8 (PNG)", LRP_PIXEL_U8_GAMMA, 4}};
  for (const auto &f : formats) {
    const size_t pbytes = (size_t)size * size * f.px_bytes;
    std::vector<void *> src(n), dst(n);
    for (int i = 0; i < n; ++i) {
      LK(lrp_host_alloc(&src[i], pbytes));
      LK(lrp_host_alloc(&dst[i], pbytes));
      if (f.fmt == LRP_PIXEL_F16) { uint16_t *h = (uint16_t *)src[i]; for (size_t k = 0; k < pbytes / 2; ++k) h[k] = (uint16_t)(0x3000u + (k * 2654435761u >> 22)); }
      else { uint8_t *b = (uint8_t *)src[i]; for (size_t k = 0; k < pbytes; ++k) b[k] = (uint8_t)(k * 2654435761u >> 24); }
      memset(dst[i], 0, pbytes);
    }
    for (int streams : {1, 3, 4}) {
      lrp_context *ctx = nullptr;
      LK(lrp_context_create(&ctx, 0, streams));
      double best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {
          in.data = (float *)src[i]; out.data = (float *)dst[i];
          LK(lrp_context_submit_packed(ctx, &in, f.fmt, 4, &out, f.fmt, 4, 255u, 1, LRP_BICUBIC, nullptr, nullptr, nullptr));
        }
        LK(lrp_context_wait(ctx));
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt < best) best = dt;
      }
      printf("packed %-20s streams=%d: %.0f Mpix/s staged, %.2f ms/frame, %.1f GB/s PCIe (both directions)\n", f.name, streams,
             (double)n * size * size / best / 1e6, best / n * 1e3, 2.0 * n * pbytes / best / 1e9);
      lrp_context_destroy(ctx);
    }
    for (int i = 0; i < n; ++i) { lrp_host_free(src[i]); lrp_host_free(dst[i]); }
  }
  return 0;
}
