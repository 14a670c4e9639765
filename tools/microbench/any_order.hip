// any_order.hip — do two kernels launched back to back on ONE stream overlap when the second carries hipExtAnyOrderLaunch
// (no barrier bit in its AQL packet)?  Each kernel keeps a quarter of the chip busy for ~T us; serial: 2T, overlapped: ~T.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/any_order.hip -o tools/microbench/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(float *out, int iters) {
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) a = a * 1.0000001f + 1e-7f;
  if (a == 12345.0f) out[blockIdx.x] = a;
}
int main() {
  float *d;
  hipMalloc(&d, 1 << 20);
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 200000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, s);
      hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, d, iters);
      if (mode == 1) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, d, iters);
      if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, iters);
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::printf("%s: %.1f us\n", mode == 0 ? "one kernel" : mode == 1 ? "two kernels, ordered" : "two kernels, second any-order", ms * 1e3f);
    }
  }
  return 0;
}
