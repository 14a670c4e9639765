// lrp_pixel_kernels.hip — pixel-format conversion on the device.
//
// The reference's codecs hand the hot path interleaved float32 buffers: read_exr widens HALF
// channels, read_png / read_jpeg apply v = pow(p / 255, 2.2) to 8-bit samples, save_exr narrows
// to HALF and save_png quantises with uint8(255.9 * pow(clamp(v, 0, 1), 1 / 2.2))
// (src/image_formats.cpp:64-66, 155-158, 196-198, 266-295, 318-333).  Doing those conversions on
// the host means 16 bytes per RGBA pixel over PCIe in each direction for data that is 8 (half) or
// 4 (8-bit) bytes on disk — and the copies, not the kernel, are what a file costs.  These kernels
// convert on the device: the frame crosses PCIe in its file format.
//
// Bit-exactness with the host codecs:
//   half <-> float   include/lrp_half.h, integer arithmetic, shared with cli/lrp_image_io.cpp;
//   8-bit decode     a 256-entry table of pow(k / 255, 2.2) filled by the HOST's powf, the
//                    very calls the codecs make;
//   8-bit encode     q(s) = uint8(255.9f * powf(s, 1 / 2.2f)) is a non-decreasing step function
//                    of s in [0, 1] (checked over all 2^30 floats of the interval in
//                    tests/test_math_vs_libm.py), so it is a search in the 255 thresholds
//                    t[k] = min { s : q(s) >= k } — found on the host with the host's powf by
//                    bisection over the float bit patterns.  No device pow is involved.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/lrp.h"
#include "../../include/lrp_half.h"

namespace lrp {

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 256 * 8;

struct PixelTables {
  float decode[256];    // pow(k / 255, 2.2)
  float threshold[256]; // threshold[k] = smallest s in [0, 1] whose code is >= k (threshold[0] = 0)
};

uint8_t quantise_host(float s) { return (uint8_t)(255.9f * std::pow(s, 1.0f / 2.2f)); } // src/image_formats.cpp:156-157

const PixelTables &host_tables() {
  static PixelTables t;
  static std::once_flag once;
  std::call_once(once, [] {
    for (int k = 0; k < 256; ++k) t.decode[k] = std::pow((float)k / 255.0f, 2.2f);
    t.threshold[0] = 0.0f;
    for (int k = 1; k < 256; ++k) {
      // non-negative floats order like their bit patterns: bisect for the first pattern whose code reaches k
      uint32_t lo = 0u, hi = 0x3f800000u; // q(0) = 0 < k <= 255 = q(1)
      while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        float s;
        std::memcpy(&s, &mid, 4);
        if (quantise_host(s) >= k)
          hi = mid;
        else
          lo = mid;
      }
      std::memcpy(&t.threshold[k], &hi, 4);
    }
  });
  return t;
}

// std::max(0.0f, std::min(1.0f, s)) with libstdc++'s comparison direction: NaN -> 1, -0 -> +0
__device__ __forceinline__ float unit_clamp_png(float v) {
  const float m = (v < 1.0f) ? v : 1.0f;
  return (0.0f < m) ? m : 0.0f;
}

__global__ __launch_bounds__(kBlock) void decode_kernel(const void *src, int format, int src_channels, float *dst,
                                                         int dst_channels, size_t n_pixels, const PixelTables *tables) {
  __shared__ float lut[256];
  if (format == LRP_PIXEL_U8_GAMMA) {
    lut[threadIdx.x] = tables->decode[threadIdx.x];
    __syncthreads();
  }
  const int copy = src_channels < dst_channels ? src_channels : dst_channels;
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < n_pixels; p += (size_t)gridDim.x * kBlock) {
    float *out = dst + p * dst_channels;
    if (format == LRP_PIXEL_F16) {
      const uint16_t *in = static_cast<const uint16_t *>(src) + p * src_channels;
      for (int c = 0; c < copy; ++c) out[c] = __uint_as_float(lrp_half_to_float_bits(in[c]));
    } else if (format == LRP_PIXEL_U8_GAMMA) {
      const uint8_t *in = static_cast<const uint8_t *>(src) + p * src_channels;
      for (int c = 0; c < copy; ++c) out[c] = lut[in[c]];
    } else {
      const float *in = static_cast<const float *>(src) + p * src_channels;
      for (int c = 0; c < copy; ++c) out[c] = in[c];
    }
    for (int c = copy; c < dst_channels; ++c) out[c] = 0.0f;
  }
}

__global__ __launch_bounds__(kBlock) void encode_kernel(const float *src, int src_channels, void *dst, int format,
                                                         int dst_channels, unsigned fill, size_t n_pixels,
                                                         const PixelTables *tables) {
  __shared__ float thr[256];
  if (format == LRP_PIXEL_U8_GAMMA) {
    thr[threadIdx.x] = tables->threshold[threadIdx.x];
    __syncthreads();
  }
  const int copy = src_channels < dst_channels ? src_channels : dst_channels;
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < n_pixels; p += (size_t)gridDim.x * kBlock) {
    const float *in = src + p * src_channels;
    if (format == LRP_PIXEL_F16) {
      uint16_t *out = static_cast<uint16_t *>(dst) + p * dst_channels;
      for (int c = 0; c < copy; ++c) out[c] = lrp_float_bits_to_half(__float_as_uint(in[c]));
      for (int c = copy; c < dst_channels; ++c) out[c] = (uint16_t)fill;
    } else if (format == LRP_PIXEL_U8_GAMMA) {
      uint8_t *out = static_cast<uint8_t *>(dst) + p * dst_channels;
      for (int c = 0; c < copy; ++c) {
        const float s = unit_clamp_png(in[c]);
        // code = number of thresholds 1..255 that s has reached (thr is non-decreasing): 8 halving steps
        int lo = 0, hi = 256; // invariant: thr[lo] <= s (thr[0] = 0), s < thr[hi] (thr[256] = +inf)
#pragma unroll
        for (int step = 0; step < 8; ++step) {
          const int mid = (lo + hi) >> 1;
          if (thr[mid] <= s)
            lo = mid;
          else
            hi = mid;
        }
        out[c] = (uint8_t)lo;
      }
      for (int c = copy; c < dst_channels; ++c) out[c] = (uint8_t)fill;
    } else {
      float *out = static_cast<float *>(dst) + p * dst_channels;
      for (int c = 0; c < copy; ++c) out[c] = in[c];
      for (int c = copy; c < dst_channels; ++c) out[c] = __uint_as_float(fill);
    }
  }
}

std::mutex g_tables_mutex;
std::vector<PixelTables *> g_device_tables; // [device], uploaded on first use, never freed (2 KiB)

hipError_t device_tables(int device, hipStream_t stream, const PixelTables **out) {
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  if (g_device_tables.size() <= (size_t)device) g_device_tables.resize((size_t)device + 1, nullptr);
  if (!g_device_tables[(size_t)device]) {
    PixelTables *d = nullptr;
    hipError_t e = hipMalloc(&d, sizeof(PixelTables));
    if (e != hipSuccess) return e;
    // the host tables are static: the copy may complete later, every kernel that reads them is behind it on a stream
    // that is ordered after this one only for the FIRST caller — so wait once here
    e = hipMemcpyAsync(d, &host_tables(), sizeof(PixelTables), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
      (void)hipFree(d);
      return e;
    }
    g_device_tables[(size_t)device] = d;
  }
  *out = g_device_tables[(size_t)device];
  return hipSuccess;
}

unsigned grid_for(size_t n) {
  size_t g = (n + kBlock - 1) / kBlock;
  if (g > (size_t)kMaxBlocks) g = kMaxBlocks;
  return (unsigned)(g < 1 ? 1 : g);
}

} // namespace

size_t pixel_bytes(int format, int channels) {
  return (size_t)channels * (format == LRP_PIXEL_F16 ? 2u : (format == LRP_PIXEL_U8_GAMMA ? 1u : 4u));
}

hipError_t launch_decode_pixels(const void *src, int format, int src_channels, float *dst, int dst_channels,
                                size_t n_pixels, int device, hipStream_t stream) {
  const PixelTables *t = nullptr;
  hipError_t e = device_tables(device, stream, &t);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(decode_kernel, dim3(grid_for(n_pixels)), dim3(kBlock), 0, stream, src, format, src_channels, dst,
                     dst_channels, n_pixels, t);
  return hipGetLastError();
}

hipError_t launch_encode_pixels(const float *src, int src_channels, void *dst, int format, int dst_channels,
                                unsigned fill, size_t n_pixels, int device, hipStream_t stream) {
  const PixelTables *t = nullptr;
  hipError_t e = device_tables(device, stream, &t);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(encode_kernel, dim3(grid_for(n_pixels)), dim3(kBlock), 0, stream, src, src_channels, dst, format,
                     dst_channels, fill, n_pixels, t);
  return hipGetLastError();
}

// Host copies of the two tables (tests, and the CLI's own host-side quantiser check).
void pixel_tables_host(float decode[256], float threshold[256]) {
  const PixelTables &t = host_tables();
  std::memcpy(decode, t.decode, sizeof(t.decode));
  std::memcpy(threshold, t.threshold, sizeof(t.threshold));
}

} // namespace lrp
