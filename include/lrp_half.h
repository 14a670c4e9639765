/* lrp_half.h — IEEE binary16 <-> binary32, round to nearest even, written with integer
 * operations only so that the host codecs (cli/lrp_image_io.cpp) and the device pixel-format
 * kernels (csrc/lrp_pixel_kernels.hip) produce the same bits for every input, NaN payloads
 * included.  This is what OpenEXR's `half` type does when the reference narrows its float
 * buffers for save_exr and widens HALF channels in read_exr (src/image_formats.cpp:266-295,
 * 318-333). */
#ifndef LRP_HALF_H
#define LRP_HALF_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define LRP_HD __host__ __device__
#else
#define LRP_HD
#endif

LRP_HD static inline uint32_t lrp_half_to_float_bits(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
  if (exp == 0) {
    if (man == 0) return sign;
    int e = -1; /* subnormal half -> normal float */
    do {
      ++e;
      man <<= 1;
    } while (!(man & 0x400u));
    return sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
  }
  if (exp == 31) return sign | 0x7f800000u | (man << 13);
  return sign | ((exp + 112u) << 23) | (man << 13);
}

LRP_HD static inline uint16_t lrp_float_bits_to_half(uint32_t x) {
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) { /* inf / NaN: the payload's top ten bits kept, bit 0 set if they are all zero — Imath's
                             software conversion (what the reference's default build, without -mf16c, runs) and numpy's;
                             the F16C instruction would set the quiet bit of a signalling NaN instead */
    const uint32_t m = (x >> 13) & 0x3ffu;
    return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? (m | (m == 0u ? 1u : 0u)) : 0u));
  }
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* rounds to inf (>= 65520) */
  if (x < 0x33000001u) return (uint16_t)sign;              /* <= 2^-25: +-0 */
  if (x < 0x38800000u) {                                   /* subnormal half */
    const int e = (int)(x >> 23);
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    const int shift = 126 - e; /* 14 .. 24 */
    const uint32_t lost = m & ((1u << shift) - 1u), half_ulp = 1u << (shift - 1);
    m >>= shift;
    if (lost > half_ulp || (lost == half_ulp && (m & 1u))) ++m;
    return (uint16_t)(sign | m);
  }
  uint32_t m = x - 0x38000000u; /* re-bias the exponent */
  const uint32_t lost = m & 0x1fffu;
  m >>= 13;
  if (lost > 0x1000u || (lost == 0x1000u && (m & 1u))) ++m;
  return (uint16_t)(sign | m);
}

#endif /* LRP_HALF_H */
