// lrp_math.h — bit-reproducible binary32 transcendental functions for the
// lens-reprojection hot path, usable from host C++ and from HIP device code.
//
// WHY THIS EXISTS
// The reference's pixel loop calls the host C library for std::sin / std::cos
// (merged into sincosf by GCC -O3), std::atan, std::atan2, std::asin
// (reference src/reproject.cpp:182,185,194,254-256,262,263).  Its output bits
// are therefore a function of that libm.  A GPU cannot call glibc, and the
// device OCML routines are different algorithms, so the kernels carry their
// own implementation of exactly the algorithms glibc 2.35 (x86-64) executes,
// operation for operation, so that every result is bit-identical to the host
// library the reference links against on this image:
//
//   * sinf / cosf / sincosf — the double-precision-polynomial routines
//     (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.c, sincosf.h,
//     s_sincosf_data.c in glibc 2.35).  x86-64 glibc selects an FMA build of
//     them by ifunc on every AVX2+FMA CPU; that build contracts each
//     `a*b + c` of the polynomials and of the fast range reduction into one
//     fused multiply-add.  The fused form is what runs on this image's hosts,
//     so it is the form restated here (every fused operation is an explicit
//     lrp_fma(); nothing is left to the compiler, which is run with
//     -ffp-contract=off).
//   * atanf, atan2f, asinf — glibc's single-precision routines
//     (s_atanf.c, e_atan2f.c, e_asinf.c); pure binary32, no ifunc variants.
//
// Operation order and constants were taken from the published algorithms and
// confirmed against the machine code and .rodata of
// /lib/x86_64-linux-gnu/libm.so.6 (Ubuntu GLIBC 2.35-0ubuntu3.11); equality
// with the live library is proven by tests/test_math_vs_libm.py (exhaustive
// 2^32 sweeps of the unary functions, structured + random pairs for atan2f).
//
// Requirements on the translation unit that includes this file:
//   -ffp-contract=off, no fast-math, IEEE divide/sqrt (hipcc default
//   -fhip-fp32-correctly-rounded-divide-sqrt), denormals preserved.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define LRP_HD __host__ __device__ __forceinline__
#else
#define LRP_HD inline
#endif

namespace lrp {

LRP_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
LRP_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
LRP_HD double lrp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
LRP_HD float lrp_sqrtf(float x) { return __builtin_sqrtf(x); }
LRP_HD float lrp_fabsf(float x) { return u2f(f2u(x) & 0x7fffffffu); }

// ---------------------------------------------------------------------------
// sinf / cosf / sincosf
// ---------------------------------------------------------------------------
namespace sc {
// __sincosf_table[0] of glibc 2.35.  table[1] differs only by the sign of the
// cosine coefficients, i.e. it yields exactly the negated cosine polynomial
// (negation commutes with every rounding), so it is applied as a sign flip.
constexpr double kHpiInv = 0x1.45f306dc9c883p+23; // 2/pi * 2^24
constexpr double kHpi = 0x1.921fb54442d18p+0;     // pi/2
constexpr double kC0 = 0x1p0;
constexpr double kC1 = -0x1.ffffffd0c621cp-2;
constexpr double kC2 = 0x1.55553e1068f19p-5;
constexpr double kC3 = -0x1.6c087e89a359dp-10;
constexpr double kC4 = 0x1.99343027bf8c3p-16;
constexpr double kS1 = -0x1.555545995a603p-3;
constexpr double kS2 = 0x1.1107605230bc4p-7;
constexpr double kS3 = -0x1.994eb3774cf24p-13;
constexpr double kPi63 = 0x1.921fb54442d18p-62; // 2^-62 * pi/2 ... (pi * 2^-63)

// sine polynomial on the reduced argument (fused form of sinf_poly, n even).
LRP_HD double sin_poly(double x, double x2) {
  double x3 = x * x2;
  double s1 = lrp_fma(kS3, x2, kS2);
  double x7 = x3 * x2;
  double s = lrp_fma(x3, kS1, x);
  return lrp_fma(s1, x7, s);
}
// cosine polynomial (fused form of sinf_poly, n odd), table[0] signs.
LRP_HD double cos_poly(double x2) {
  double x4 = x2 * x2;
  double c1 = lrp_fma(x2, kC1, kC0);
  double c2 = lrp_fma(x2, kC4, kC3);
  double x6 = x4 * x2;
  double c = lrp_fma(x4, kC2, c1);
  return lrp_fma(c2, x6, c);
}

// reduce_fast: |x| < 120.  n = round(x * 2/pi), returns x - n*pi/2 (fused).
LRP_HD double reduce_fast(double x, int &n) {
  double r = x * kHpiInv;
  n = ((int32_t)r + 0x800000) >> 24;
  return lrp_fma(-(double)n, kHpi, x);
}

// reduce_large: 120 <= |x| < inf, fixed-point multiply by 4/pi bits.
LRP_HD double reduce_large(uint32_t xi, int &n) {
  // __inv_pio4[24]
  const uint32_t inv_pio4[24] = {
      0xa2u,       0xa2f9u,     0xa2f983u,   0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u,
      0x6e4e4415u, 0x4e441529u, 0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u,
      0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u,
      0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
  const int idx = (xi >> 26) & 15;
  const int shift = (xi >> 23) & 7;
  xi = (xi & 0xffffffu) | 0x800000u;
  xi <<= shift;
  uint64_t res0 = (uint32_t)(xi * inv_pio4[idx]);
  uint64_t res1 = (uint64_t)xi * inv_pio4[idx + 4];
  uint64_t res2 = (uint64_t)xi * inv_pio4[idx + 8];
  res0 = (res2 >> 32) | (res0 << 32);
  res0 += res1;
  uint64_t nn = (res0 + (1ULL << 61)) >> 62;
  res0 -= nn << 62;
  double x = (double)(int64_t)res0;
  n = (int)nn;
  return x * kPi63;
}

LRP_HD uint32_t abstop12(uint32_t bits) { return (bits >> 20) & 0x7ffu; }
} // namespace sc

// One routine yields both results with exactly the operations the three glibc
// entry points perform (they share reduction and polynomials; each output
// depends only on its own polynomial chain).
template <bool WantSin, bool WantCos>
LRP_HD void sincos_core(float y, float &sn, float &cs) {
  using namespace sc;
  const uint32_t bits = f2u(y);
  const uint32_t top = abstop12(bits);
  double x = (double)y;
  if (top < 0x3f4u) { // |y| < pi/4
    double x2 = x * x;
    if (top < 0x398u) { // |y| < 2^-12
      if (WantSin) sn = y;
      if (WantCos) cs = 1.0f;
      return;
    }
    if (WantSin) sn = (float)sin_poly(x, x2);
    if (WantCos) cs = (float)cos_poly(x2);
    return;
  }
  int n;
  uint32_t q; // quadrant selector used for sign[] and table choice
  if (top < 0x42fu) { // |y| < 120
    x = reduce_fast(x, n);
    q = (uint32_t)n;
  } else if (top < 0x7f8u) { // finite
    x = reduce_large(bits, n);
    q = (uint32_t)n + (bits >> 31);
  } else { // inf or NaN -> NaN (glibc: __math_invalidf)
    float r = (y - y) / (y - y);
    if (WantSin) sn = r;
    if (WantCos) cs = r;
    return;
  }
  // sign[] = {1,-1,-1,1}
  const double sgn = ((q + 1u) & 2u) ? -1.0 : 1.0;
  const bool neg_cos = (q & 2u) != 0; // table[1]
  const double xs = x * sgn;
  const double x2 = x * x;
  // quadrant parity swaps the roles of the two polynomials.
  if (n & 1) {
    if (WantSin) {
      double c = cos_poly(x2);
      sn = (float)(neg_cos ? -c : c);
    }
    if (WantCos) cs = (float)sin_poly(xs, x2);
  } else {
    if (WantSin) sn = (float)sin_poly(xs, x2);
    if (WantCos) {
      double c = cos_poly(x2);
      cs = (float)(neg_cos ? -c : c);
    }
  }
}

LRP_HD float sinf_(float y) {
  float s, c;
  sincos_core<true, false>(y, s, c);
  return s;
}
// cosf(y): glibc evaluates sinf_poly(x*s, x2, p, n ^ 1); the sign index for the
// sine polynomial there is sign[n & 3] with the *unshifted* n, which is the
// same table entry sincos_core uses for its cosine output.
LRP_HD float cosf_(float y) {
  float s, c;
  sincos_core<false, true>(y, s, c);
  return c;
}
LRP_HD void sincosf_(float y, float &s, float &c) { sincos_core<true, true>(y, s, c); }

// ---------------------------------------------------------------------------
// atanf  (glibc 2.35 sysdeps/ieee754/flt-32/s_atanf.c)
// ---------------------------------------------------------------------------
LRP_HD float atanf_(float x) {
  const float atanhi0 = u2f(0x3eed6338u), atanlo0 = u2f(0x31ac3769u);
  const float atanhi1 = u2f(0x3f490fdau), atanlo1 = u2f(0x33222168u);
  const float atanhi2 = u2f(0x3f7b985eu), atanlo2 = u2f(0x33140fb4u);
  const float atanhi3 = u2f(0x3fc90fdau), atanlo3 = u2f(0x33a22168u);
  const float aT0 = u2f(0x3eaaaaabu), aT1 = u2f(0xbe4ccccdu), aT2 = u2f(0x3e124925u),
              aT3 = u2f(0xbde38e38u), aT4 = u2f(0x3dba2e6eu), aT5 = u2f(0xbd9d8795u),
              aT6 = u2f(0x3d886b35u), aT7 = u2f(0xbd6ef16bu), aT8 = u2f(0x3d4bda59u),
              aT9 = u2f(0xbd15a221u), aT10 = u2f(0x3c8569d7u);
  const uint32_t hx = f2u(x);
  const uint32_t ix = hx & 0x7fffffffu;
  float hi = 0.0f, lo = 0.0f;
  bool direct = false; // id < 0
  if (ix >= 0x4c000000u) { // |x| >= 2^25
    if (ix > 0x7f800000u) return x + x; // NaN
    if ((int32_t)hx > 0) return atanlo3 + atanhi3;
    return -atanhi3 - atanlo3;
  }
  if (ix < 0x3ee00000u) {   // |x| < 0.4375
    if (ix < 0x31000000u) { // |x| < 2^-29: huge + x > one always holds
      return x;
    }
    direct = true;
  } else {
    // The four argument reductions differ only in numerator and denominator:
    // select those, then divide once (one IEEE divide per call instead of one
    // per range; the quotient is the same operation on the same operands).
    const float ax = u2f(ix);
    float num, den;
    if (ix < 0x3f980000u) {   // |x| < 1.1875
      if (ix < 0x3f300000u) { // 7/16 <= |x| < 11/16
        num = (ax + ax) - 1.0f;
        den = ax + 2.0f;
        hi = atanhi0;
        lo = atanlo0;
      } else { // 11/16 <= |x| < 19/16
        num = ax - 1.0f;
        den = ax + 1.0f;
        hi = atanhi1;
        lo = atanlo1;
      }
    } else {
      if (ix < 0x401c0000u) { // |x| < 2.4375
        num = ax - 1.5f;
        den = ax * 1.5f + 1.0f;
        hi = atanhi2;
        lo = atanlo2;
      } else { // 2.4375 <= |x| < 2^25
        num = -1.0f;
        den = ax;
        hi = atanhi3;
        lo = atanlo3;
      }
    }
    x = num / den;
  }
  const float z = x * x;
  const float w = z * z;
  // odd and even halves of the polynomial, Horner in w
  const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  const float xs = x * (s1 + s2);
  if (direct) return x - xs;
  const float r = hi - ((xs - lo) - x);
  return ((int32_t)hx < 0) ? -r : r;
}

// ---------------------------------------------------------------------------
// atan2f  (glibc 2.35 sysdeps/ieee754/flt-32/e_atan2f.c)
// ---------------------------------------------------------------------------
LRP_HD float atan2f_(float y, float x) {
  const float tiny = u2f(0x0da24260u); // 1.0e-30
  const float pi_o_4 = u2f(0x3f490fdbu);
  const float pi_o_2 = u2f(0x3fc90fdbu);
  const float pi = u2f(0x40490fdbu);
  const float pi_lo = u2f(0xb3bbbd2eu); // -8.7422776573e-08
  const uint32_t hx = f2u(x), hy = f2u(y);
  const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
  if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y; // NaN
  if (hx == 0x3f800000u) return atanf_(y);                // x == 1.0
  const uint32_t m = ((hy >> 31) & 1u) | ((hx >> 30) & 2u);
  if (iy == 0) { // y == 0
    switch (m) {
    case 0:
    case 1: return y;
    case 2: return pi + tiny;
    default: return -pi - tiny;
    }
  }
  if (ix == 0) return ((int32_t)hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000u) {
    if (iy == 0x7f800000u) {
      switch (m) {
      case 0: return pi_o_4 + tiny;
      case 1: return -pi_o_4 - tiny;
      case 2: return 3.0f * pi_o_4 + tiny;
      default: return -3.0f * pi_o_4 - tiny;
      }
    } else {
      switch (m) {
      case 0: return 0.0f;
      case 1: return -0.0f;
      case 2: return pi + tiny;
      default: return -pi - tiny;
      }
    }
  }
  if (iy == 0x7f800000u) return ((int32_t)hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int32_t d = (int32_t)(iy - ix);
  const int32_t k = d >> 23;
  float z;
  if (k > 60) {
    z = pi_o_2 + 0.5f * pi_lo; // |y/x| > 2^60
  } else if ((int32_t)hx < 0 && k < -60) {
    z = 0.0f; // |y|/x < -2^60
  } else {
    z = atanf_(lrp_fabsf(y / x));
  }
  switch (m) {
  case 0: return z;
  case 1: return -z;
  case 2: return pi - (z - pi_lo);
  default: return (z - pi_lo) - pi;
  }
}

// ---------------------------------------------------------------------------
// asinf  (glibc 2.35 sysdeps/ieee754/flt-32/e_asinf.c)
// ---------------------------------------------------------------------------
LRP_HD float asinf_(float x) {
  const float pio2_hi = u2f(0x3fc90fdbu);
  const float pio2_lo = u2f(0xb33bbd2eu);
  const float pio4_hi = u2f(0x3f490fdbu);
  const float p0 = u2f(0x3e2aaae4u), p1 = u2f(0x3d9980f2u), p2 = u2f(0x3d3a3f25u),
              p3 = u2f(0x3cc6141eu), p4 = u2f(0x3d2cb694u);
  const uint32_t hx = f2u(x);
  const uint32_t ix = hx & 0x7fffffffu;
  if (ix == 0x3f800000u) return x * pio2_hi + x * pio2_lo; // |x| == 1
  if (ix > 0x3f800000u) return (x - x) / (x - x);          // |x| > 1 or NaN
  if (ix < 0x3f000000u) {                                   // |x| < 0.5
    if (ix < 0x32000000u) return x;                         // |x| < 2^-27
    const float t = x * x;
    const float w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    return x + x * w;
  }
  // 0.5 <= |x| < 1
  float w = 1.0f - u2f(ix);
  float t = w * 0.5f;
  float p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
  const float s = lrp_sqrtf(t);
  if (ix >= 0x3f79999au) { // |x| > 0.975
    t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
  } else {
    w = u2f(f2u(s) & 0xfffff000u);
    const float c = (t - w * w) / (s + w);
    const float r = p;
    p = 2.0f * s * r - (pio2_lo - 2.0f * c);
    const float q = pio4_hi - 2.0f * w;
    t = pio4_hi - (p - q);
  }
  return ((int32_t)hx > 0) ? t : -t;
}

} // namespace lrp
