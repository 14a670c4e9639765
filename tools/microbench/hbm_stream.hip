// hbm_stream.hip — what the HBM interface of this box delivers to plain streaming kernels (the yardstick DESIGN.md reads the
// measured traffic of the reprojection kernels against): device-to-device hipMemcpy, a float4 copy kernel, a read-only sum and a
// write-only fill over 1 GiB, non-temporal and default cache policy.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/hbm_stream.hip -o tools/microbench/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT> __global__ __launch_bounds__(256) void copy_k(const v4f *a, v4f *b, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) {
    const v4f v = NT ? __builtin_nontemporal_load(a + i) : a[i];
    if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
  }
}
template <bool NT> __global__ __launch_bounds__(256) void fill_k(v4f *b, size_t n) {
  const v4f v = {1.0f, 2.0f, 3.0f, 4.0f};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) {
    if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
  }
}
template <bool NT> __global__ __launch_bounds__(256) void sum_k(const v4f *a, float *out, size_t n) {
  v4f s = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) s += NT ? __builtin_nontemporal_load(a + i) : a[i];
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}
int main() {
  const size_t bytes = 1ull << 30, n = bytes / 16;
  v4f *a, *b; float *o;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 64);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char *name, double moved, auto fn) {
    float best = 1e30f;
    for (int r = 0; r < 6; ++r) { hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms; }
    std::printf("%-44s %8.1f us  %6.2f TB/s\n", name, best * 1e3, moved / (best * 1e-3) / 1e12);
  };
  time("hipMemcpy device to device (1 GiB + 1 GiB)", 2.0 * bytes, [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
  for (int grid : {2048, 8192, 32768}) {
    char nm[96];
    std::snprintf(nm, sizeof nm, "copy kernel, grid %d, default policy", grid); time(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_k<false>, dim3(grid), dim3(256), 0, 0, a, b, n); });
    std::snprintf(nm, sizeof nm, "copy kernel, grid %d, non-temporal", grid); time(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_k<true>, dim3(grid), dim3(256), 0, 0, a, b, n); });
    std::snprintf(nm, sizeof nm, "fill kernel, grid %d, non-temporal", grid); time(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL(fill_k<true>, dim3(grid), dim3(256), 0, 0, b, n); });
    std::snprintf(nm, sizeof nm, "sum kernel (reads), grid %d, non-temporal", grid); time(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL(sum_k<true>, dim3(grid), dim3(256), 0, 0, a, o, n); });
    std::snprintf(nm, sizeof nm, "sum kernel (reads), grid %d, default", grid); time(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL(sum_k<false>, dim3(grid), dim3(256), 0, 0, a, o, n); });
  }
  return 0;
}
