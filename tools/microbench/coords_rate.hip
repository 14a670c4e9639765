// coords_rate.hip — cost of the source-coordinate math of the tile kernel
// (lrp_kernel_v2.h pixel_source) on register data, per lens pair, as a function
// of resident wavefronts per SIMD and of pixels in flight per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../image-lens-reproject_amd/csrc coords_rate.hip -o coords_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "lrp_kernel_v2.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace lrp;
constexpr int ITERS = 256;

template <int OutLens, int InMode, int PIX> __global__ __launch_bounds__(256) void k(const KParams P, float *out, int lds_pad) {
  extern __shared__ float pad[];
  if (lds_pad < 0) pad[threadIdx.x] = 1.0f;
  ColTerms col{(threadIdx.x & 63) * 3e-4f - 0.01f, 0.3f};
  float acc = 0.0f;
  for (int it = 0; it < ITERS; ++it) {
    float sx[PIX], sy[PIX];
#pragma unroll
    for (int p = 0; p < PIX; ++p) {
      ColTerms c = col;
      c.a += p * 1e-3f;
      pixel_source<OutLens, InMode>(P, c, (it * PIX + p) & 1023, 0, sx[p], sy[p]);
    }
#pragma unroll
    for (int p = 0; p < PIX; ++p) acc += sx[p] + sy[p];
    col.a += 1e-5f;
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int OutLens, int InMode, int PIX> int run(const char *name, const KParams &P, int waves_per_simd) {
  float *out; CHECK(hipMalloc(&out, 4));
  const int lds = (160 * 1024) / waves_per_simd - 512;
  CHECK(hipFuncSetAttribute((const void *)k<OutLens, InMode, PIX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  const int blocks = 256 * waves_per_simd * 4;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<OutLens, InMode, PIX>), dim3(blocks), dim3(256), lds, 0, P, out, 0);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<OutLens, InMode, PIX>), dim3(blocks), dim3(256), lds, 0, P, out, 0);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double passes_per_simd = (double)blocks * 4 * ITERS * PIX / 1024.0;
  printf("%-14s pix/lane %d waves/SIMD %d: %7.1f cycles per wave-pass per SIMD (@2.4GHz) = %6.1f us per 4K frame\n", name, PIX,
         waves_per_simd, best * 1e-3 * 2.4e9 / passes_per_simd, 256.0 * (best * 1e-3 / passes_per_simd) * 1e6);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  KParams P; memset(&P, 0, sizeof(P));
  float *tab; CHECK(hipMalloc(&tab, 4096 * 4));
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (i - 512) * 9.7e-4f;
  CHECK(hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice));
  P.in_w = P.in_h = P.out_w = P.out_h = 4096; P.num_samples = 1; P.col_tab = tab; P.row_tab = tab;
  P.in_lens.p[0] = 3.14159265f; P.in_lens.sensor_width = 36; P.in_lens.sensor_height = 36; P.in_focal = 36 / 3.14159265f;
  KParams Q = P; // equirect source
  Q.in_lens.p[0] = -1.5707964f; Q.in_lens.p[1] = 1.5707964f; Q.in_lens.p[2] = -3.1415927f; Q.in_lens.p[3] = 3.1415927f;
  Q.in_lon_span = 6.2831855f; Q.in_lat_span = 3.1415927f;
  for (int w : {1, 2, 3, 4, 8}) {
    run<kRect, kInEquidistant, 1>("rect<-eqd", P, w);
    run<kRect, kInEquidistant, 4>("rect<-eqd", P, w);
    run<kRect, kInEquirectLoop, 1>("rect<-eqr", Q, w);
    run<kRect, kInEquirectLoop, 4>("rect<-eqr", Q, w);
  }
  return 0;
}
