// lrp_aux_kernels.hip — small streaming kernels around the hot path:
// stand-alone post_process, synthetic frame generation, device math probe.
#include <hip/hip_runtime.h>

#include "lrp_device.h"

namespace lrp {

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 256 * 8; // 256 CUs x 8 resident workgroups, grid-stride beyond

// post_process (src/reproject.cpp:421-437), in place, first min(C,3) channels.
__global__ __launch_bounds__(kBlock) void post_process_kernel(float *data, uint32_t n_pixels, int channels,
                                                               float exposure, float reinhard) {
  const int ch = channels < 3 ? channels : 3;
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n_pixels; p += gridDim.x * kBlock) {
    float *px = data + (size_t)p * channels;
    if (channels == 4) {
      float4 q = *reinterpret_cast<float4 *>(px);
      q.x = tonemap(q.x, exposure, reinhard);
      q.y = tonemap(q.y, exposure, reinhard);
      q.z = tonemap(q.z, exposure, reinhard);
      *reinterpret_cast<float4 *>(px) = q;
    } else {
      for (int c = 0; c < ch; ++c) px[c] = tonemap(px[c], exposure, reinhard);
    }
  }
}

__device__ __forceinline__ uint32_t mix32(uint32_t seed, uint32_t index) {
  uint32_t h = index * 0x9E3779B9u + seed;
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}

__device__ __forceinline__ float synth_value(uint32_t seed, uint32_t index, bool depth) {
  const uint32_t h = mix32(seed, index);
  const float u = (float)(h >> 21) * (1.0f / 2048.0f);
  if (!depth) return u;
  const float d = 0.1f + u * 99.9f;
  uint32_t b = f2u(d);
  b = (b + 0x00000FFFu + ((b >> 13) & 1u)) & 0xFFFFE000u;
  return u2f(b);
}

__global__ __launch_bounds__(kBlock) void synth_fill_kernel(float *data, uint32_t n_elems, int channels,
                                                             uint32_t seed, int depth_channel) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_elems; i += gridDim.x * kBlock) {
    const int c = (int)(i % (uint32_t)channels);
    data[i] = synth_value(seed, i, c == depth_channel);
  }
}

__global__ __launch_bounds__(kBlock) void math_eval_kernel(int func, const float *a, const float *b, float *out,
                                                            size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const float x = a[i];
    const float y = b ? b[i] : 0.0f;
    float r, t;
    switch (func) {
    case 0: r = sinf_(x); break;
    case 1: r = cosf_(x); break;
    case 2: sincosf_(x, r, t); break;
    case 3: sincosf_(x, t, r); break;
    case 4: r = atanf_(x); break;
    case 5: r = asinf_(x); break;
    case 6: r = atan2f_(x, y); break;
    case 7: r = x / y; break;
    case 8: r = lrp_sqrtf(x); break;
    case 9: r = (float)trunc_x86(x); break;
    default: r = 0.0f; break;
    }
    out[i] = r;
  }
}

// Order-independent 64-bit checksum of a float buffer's bit patterns: the sum (mod 2^64) over all
// elements of a 64-bit hash of (bits, index).  Any association of the additions gives the same
// value, so the grid shape does not matter; tests/bench compare per-image values across runs,
// GPU counts and against the same sum evaluated on the host.
__device__ __forceinline__ unsigned long long checksum_term(uint32_t bits, uint32_t index) {
  const uint32_t lo = mix32(bits, index);
  const uint32_t hi = mix32(bits ^ 0xA5A5A5A5u, index * 2u + 0x7F4A7C15u);
  return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(kBlock) void checksum_kernel(const uint32_t *data, size_t n, unsigned long long *out) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    acc += checksum_term(data[i], (uint32_t)i);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63u) == 0) atomicAdd(out, acc);
}

inline unsigned grid_for(size_t n) {
  size_t g = (n + kBlock - 1) / kBlock;
  if (g > (size_t)kMaxBlocks) g = kMaxBlocks;
  if (g < 1) g = 1;
  return (unsigned)g;
}

} // namespace

hipError_t launch_post_process(float *data, uint32_t n_pixels, int channels, float exposure, float reinhard,
                               hipStream_t stream) {
  hipLaunchKernelGGL(post_process_kernel, dim3(grid_for(n_pixels)), dim3(kBlock), 0, stream, data, n_pixels,
                     channels, exposure, reinhard);
  return hipGetLastError();
}

hipError_t launch_synth_fill(float *data, uint32_t n_elems, int channels, uint32_t seed, int depth_channel,
                             hipStream_t stream) {
  hipLaunchKernelGGL(synth_fill_kernel, dim3(grid_for(n_elems)), dim3(kBlock), 0, stream, data, n_elems, channels,
                     seed, depth_channel);
  return hipGetLastError();
}

hipError_t launch_checksum(const float *data, size_t n, unsigned long long *out, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned long long), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(checksum_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, reinterpret_cast<const uint32_t *>(data), n,
                     out);
  return hipGetLastError();
}

hipError_t launch_math_eval(int func, const float *a, const float *b, float *out, size_t n, hipStream_t stream) {
  hipLaunchKernelGGL(math_eval_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, func, a, b, out, n);
  return hipGetLastError();
}

} // namespace lrp
