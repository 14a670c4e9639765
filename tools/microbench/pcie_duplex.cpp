// pcie_duplex.cpp — can H2D and D2H copies overlap on this platform?  Pinned host
// buffers, two streams.  Build: hipcc -O2 pcie_duplex.cpp -o pcie_duplex
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  const size_t n = 256u << 20;
  void *h_in, *h_out, *d_a, *d_b;
  CK(hipHostMalloc(&h_in, n, hipHostMallocDefault)); CK(hipHostMalloc(&h_out, n, hipHostMallocDefault));
  CK(hipMalloc(&d_a, n)); CK(hipMalloc(&d_b, n));
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
  for (int rep = 0; rep < 2; ++rep) {
    auto t0 = now();
    for (int i = 0; i < 8; ++i) CK(hipMemcpyAsync(d_a, h_in, n, hipMemcpyHostToDevice, s0));
    CK(hipStreamSynchronize(s0));
    auto t1 = now();
    for (int i = 0; i < 8; ++i) CK(hipMemcpyAsync(h_out, d_b, n, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s1));
    auto t2 = now();
    for (int i = 0; i < 8; ++i) { CK(hipMemcpyAsync(d_a, h_in, n, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(h_out, d_b, n, hipMemcpyDeviceToHost, s1)); }
    CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
    auto t3 = now();
    printf("H2D alone %.1f GB/s, D2H alone %.1f GB/s, both together %.1f GB/s aggregate\n", 8 * n / secs(t0, t1) / 1e9,
           8 * n / secs(t1, t2) / 1e9, 16 * n / secs(t2, t3) / 1e9);
  }
  return 0;
}
