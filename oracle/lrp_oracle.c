/*
 * lrp_oracle.c — CPU restatement of the reference hot path (see lrp_oracle.h:
 * TEST INFRASTRUCTURE, parity unpinned).  Citations are file:line of
 * /root/reference (IDLabMedia/image-lens-reproject @ v1).
 *
 * Build: gcc -std=c11 -O3 -ffp-contract=off (no -march, no fast-math), i.e. the
 * reference's Release flags (CMakeLists.txt:8,64: C++17, default -O3 -DNDEBUG).
 * x86-64 baseline has no FMA, so every multiply and add below rounds on its own,
 * exactly like the reference object code.
 */
#include "lrp_oracle.h"

#include <limits.h>
#include <math.h>
#include <stddef.h>

/* ---- scalar semantics the reference inherits from x86-64 / libstdc++ ------ */

/* int(float) as the reference's cvttss2si executes it: truncation toward zero;
 * NaN, +-inf and anything outside int32 give INT_MIN ("integer indefinite").
 * Used wherever the reference writes int(sx + k) (src/reproject.cpp:43-47,
 * 60-67, 114-127). */
static int trunc_x86(float v) {
  if (!(fabsf(v) < 2147483648.0f)) return INT_MIN;
  return (int)v;
}

/* clamp<int> = max(lo, min(hi, x))  (src/reproject.cpp:33-35). */
static int clamp_index(int x, int lo, int hi) {
  int m = (hi < x) ? hi : x; /* std::min(hi, x) */
  return (lo < m) ? m : lo;  /* std::max(lo, m) */
}

/* Horizontal wrap (int(..) + W) % W of the LoopHorizontally samplers
 * (src/reproject.cpp:43, 60-61, 114-117): two's-complement add, C remainder.
 * When the remainder is negative the reference indexes out of bounds
 * (undefined behaviour; only reachable with non-finite / absurd coordinates
 * and a non-power-of-two width).  This restatement — and the HIP path — define
 * that case as column 0. */
static int wrap_index(int i, int w) {
  int t = (int)((unsigned)i + (unsigned)w);
  int r = t % w;
  return (r < 0) ? 0 : r;
}

/* std::max(0.0f, std::min(1.0f, v)) with libstdc++'s comparison direction
 * (src/reproject.cpp:70-71, 130-131): NaN -> 1.0f, -0.0f -> +0.0f. */
static float unit_clamp(float v) {
  float m = (v < 1.0f) ? v : 1.0f;
  return (0.0f < m) ? m : 0.0f;
}

/* ---- lens models ---------------------------------------------------------- */

/* Output pixel (centred coordinates) -> ray.  Restates rectilinear_to_vec
 * (src/reproject.cpp:152-158), equidistant_to_vec (:171-186) and
 * equirectangular_to_vec (:245-257). */
static void target_ray(const lrpo_lens *L, float img_w, float img_h, float cx, float cy, float v[3]) {
  switch (L->type) {
  case LRPO_RECTILINEAR: {
    float focal = L->u[0];
    v[0] = cx / img_w * L->sensor_width / focal;
    v[1] = cy / img_h * L->sensor_height / focal;
    v[2] = -1.0f;
    break;
  }
  case LRPO_FISHEYE_EQUIDISTANT: {
    float fov = L->u[0];
    float r_px = sqrtf(cx * cx + cy * cy);
    float r_mm = r_px / img_w * L->sensor_width;
    float focal = L->sensor_width / fov;
    float theta = r_mm / focal;
    float s = sinf(theta) / r_px;
    v[0] = s * cx;
    v[1] = s * cy;
    v[2] = cosf(theta); /* +cos: the reference's sign quirk, kept */
    break;
  }
  default: { /* LRPO_EQUIRECTANGULAR */
    float lat_min = L->u[0], lat_max = L->u[1], lon_min = L->u[2], lon_max = L->u[3];
    float lon_span = lon_max - lon_min;
    float lat_span = lat_max - lat_min;
    float lon = ((cx / img_w) + 0.5f) * lon_span + lon_min;
    float lat = ((cy / img_h) + 0.5f) * lat_span + lat_min;
    v[0] = sinf(lon);
    v[2] = -cosf(lon);
    v[1] = sinf(lat); /* un-normalised on purpose (no cos(lat) factor) */
    break;
  }
  }
}

/* Ray -> centred source coordinates.  Restates vec_to_rectilinear
 * (src/reproject.cpp:160-167), vec_to_equidistant (:188-206) and
 * vec_to_equirectangular (:259-271). */
static void ray_to_source(const lrpo_lens *L, float img_w, float img_h, float x, float y, float z,
                          float *cx, float *cy) {
  switch (L->type) {
  case LRPO_RECTILINEAR: {
    float focal = L->u[0];
    x /= -z;
    y /= -z;
    *cx = x * img_w / L->sensor_width * focal;
    *cy = y * img_h / L->sensor_height * focal;
    break;
  }
  case LRPO_FISHEYE_EQUIDISTANT: {
    float fov = L->u[0];
    x /= -z;
    y /= -z;
    float r = sqrtf(x * x + y * y);
    float theta = atanf(r);
    float focal = L->sensor_width / fov;
    float r_mm = focal * theta;
    float r_px = r_mm / L->sensor_width * img_w;
    *cx = x / r * r_px;
    *cy = y / r * r_px;
    break;
  }
  default: { /* LRPO_EQUIRECTANGULAR */
    float lat_min = L->u[0], lat_max = L->u[1], lon_min = L->u[2], lon_max = L->u[3];
    float theta = -atan2f(-x, -z);
    float phi = asinf(y / sqrtf(x * x + y * y + z * z));
    float lon_span = lon_max - lon_min;
    float lat_span = lat_max - lat_min;
    *cx = ((theta - lon_min) / lon_span - 0.5f) * img_w;
    *cy = ((phi - lat_min) / lat_span - 0.5f) * img_h;
    break;
  }
  }
}

/* ---- samplers ------------------------------------------------------------- */

static int column(int i, int w, int loop) { return loop ? wrap_index(i, w) : clamp_index(i, 0, w - 1); }

/* sample_nearest (src/reproject.cpp:39-53). */
static void tap_nearest(const lrpo_image *img, int loop, float sx, float sy, float *out) {
  int lx = column(trunc_x86(sx + 0.5f), img->width, loop);
  int ly = clamp_index(trunc_x86(sy + 0.5f), 0, img->height - 1);
  int pitch = img->width * img->channels;
  const float *src = img->data + (ptrdiff_t)ly * pitch + (ptrdiff_t)lx * img->channels;
  for (int c = 0; c < img->channels; ++c) out[c] = src[c];
}

/* sample_bilinear (src/reproject.cpp:55-90): horizontal lerps, then vertical;
 * weights from the already wrapped/clamped lower index. */
static void tap_bilinear(const lrpo_image *img, int loop, float sx, float sy, float *out) {
  int w = img->width, h = img->height, C = img->channels;
  int lx = column(trunc_x86(sx), w, loop);
  int ux = column(trunc_x86(sx + 1.0f), w, loop);
  int ly = clamp_index(trunc_x86(sy), 0, h - 1);
  int uy = clamp_index(trunc_x86(sy + 1.0f), 0, h - 1);
  float fx = unit_clamp(sx - (float)lx);
  float fy = unit_clamp(sy - (float)ly);
  float cfx = 1.0f - fx;
  float cfy = 1.0f - fy;
  ptrdiff_t pitch = (ptrdiff_t)w * C;
  const float *row_l = img->data + ly * pitch;
  const float *row_u = img->data + uy * pitch;
  for (int c = 0; c < C; ++c) {
    float ll = row_l[lx * C + c];
    float lu = row_l[ux * C + c];
    float ul = row_u[lx * C + c];
    float uu = row_u[ux * C + c];
    float lo = fx * lu + cfx * ll;
    float hi = fx * uu + cfx * ul;
    out[c] = fy * hi + cfy * lo;
  }
}

/* cubicInterpolate (src/reproject.cpp:92-98): Catmull-Rom in the reference's
 * exact association order. */
static float catmull_rom(float a, float b, float c, float d, float t) {
  float inner = ((3.0f * (b - c)) + d) - a;
  float mid = ((((2.0f * a) - (5.0f * b)) + (4.0f * c)) - d) + t * inner;
  float outer = (c - a) + t * mid;
  return b + (0.5f * t) * outer;
}

/* sample_bicubic + bicubicInterpolate (src/reproject.cpp:100-148): four
 * independently truncated tap columns/rows, vertical cubic per column first,
 * then one horizontal cubic. */
static void tap_bicubic(const lrpo_image *img, int loop, float sx, float sy, float *out) {
  int w = img->width, h = img->height, C = img->channels;
  int xs[4], ys[4];
  xs[0] = column(trunc_x86(sx - 1.0f), w, loop);
  xs[1] = column(trunc_x86(sx), w, loop);
  xs[2] = column(trunc_x86(sx + 1.0f), w, loop);
  xs[3] = column(trunc_x86(sx + 2.0f), w, loop);
  ys[0] = clamp_index(trunc_x86(sy - 1.0f), 0, h - 1);
  ys[1] = clamp_index(trunc_x86(sy), 0, h - 1);
  ys[2] = clamp_index(trunc_x86(sy + 1.0f), 0, h - 1);
  ys[3] = clamp_index(trunc_x86(sy + 2.0f), 0, h - 1);
  float fx = unit_clamp(sx - (float)xs[1]);
  float fy = unit_clamp(sy - (float)ys[1]);
  ptrdiff_t pitch = (ptrdiff_t)w * C;
  for (int c = 0; c < C; ++c) {
    float col[4];
    for (int i = 0; i < 4; ++i) {
      const float *p = img->data + (ptrdiff_t)xs[i] * C + c;
      col[i] = catmull_rom(p[ys[0] * pitch], p[ys[1] * pitch], p[ys[2] * pitch], p[ys[3] * pitch], fy);
    }
    out[c] = catmull_rom(col[0], col[1], col[2], col[3], fx);
  }
}

/* ---- dispatch + pixel loop ------------------------------------------------ */

static int lens_supported(int type) {
  return type == LRPO_RECTILINEAR || type == LRPO_FISHEYE_EQUIDISTANT || type == LRPO_EQUIRECTANGULAR;
}

/* LoopHorizontally decision of reproject_to (src/reproject.cpp:386-394): float
 * span, double comparison against 2*M_PI, float threshold. */
static int source_wraps(const lrpo_lens *L) {
  if (L->type != LRPO_EQUIRECTANGULAR) return 0;
  float long_range = L->u[3] - L->u[2];
  return fabs((double)long_range - (2 * M_PI)) < 1e-5f;
}

static int check_dispatch(const lrpo_image *in, const lrpo_image *out, int interpolation) {
  /* order of the reference's tests: output lens (src/reproject.cpp:408-418),
   * input lens (:378-398), interpolation (:352-367). */
  if (!lens_supported(out->lens.type)) return LRPO_ERR_OUTPUT_LENS;
  if (!lens_supported(in->lens.type)) return LRPO_ERR_INPUT_LENS;
  if (interpolation != LRPO_NEAREST && interpolation != LRPO_BILINEAR && interpolation != LRPO_BICUBIC)
    return LRPO_ERR_INTERPOLATION;
  return LRPO_OK;
}

/* One sub-sample: output-lens ray, optional rotation, input-lens projection,
 * shift to top-left-origin texel coordinates (src/reproject.cpp:300-324). */
static void source_position(const lrpo_image *in, const lrpo_image *out, const float *rm, float scx,
                            float scy, float *sx, float *sy) {
  float v[3];
  target_ray(&out->lens, (float)out->width, (float)out->height, scx, scy, v);
  if (rm) {
    float nx = rm[0] * v[0] + rm[1] * v[1] + rm[2] * v[2];
    float ny = rm[3] * v[0] + rm[4] * v[1] + rm[5] * v[2];
    float nz = rm[6] * v[0] + rm[7] * v[1] + rm[8] * v[2];
    v[0] = nx;
    v[1] = ny;
    v[2] = nz;
  }
  float px, py;
  ray_to_source(&in->lens, (float)in->width, (float)in->height, v[0], v[1], v[2], &px, &py);
  *sx = (px - 0.5f) + in->width * 0.5f;
  *sy = (py - 0.5f) + in->height * 0.5f;
}

#define LRPO_MAX_STACK_CHANNELS 64 /* the reference allocates 2 * C floats (src/reproject.cpp:281) */

int lrpo_reproject_rows(const lrpo_image *in, lrpo_image *out, int num_samples, int interpolation,
                        const float *rotation, int y_begin, int y_end) {
  int rc = check_dispatch(in, out, interpolation);
  if (rc != LRPO_OK) return rc;
  const int loop = source_wraps(&in->lens);
  const int C = out->channels;
  const ptrdiff_t pitch = (ptrdiff_t)out->width * C;
  /* src/reproject.cpp:280: int product, then one float divide. */
  const float normalize = 1.0f / (num_samples * num_samples);
  float stack_buf[2 * LRPO_MAX_STACK_CHANNELS];
  float *acc = stack_buf, *tap = stack_buf + LRPO_MAX_STACK_CHANNELS;
  if (C > LRPO_MAX_STACK_CHANNELS) return -1;
  if (y_begin < 0) y_begin = 0;
  if (y_end > out->height) y_end = out->height;

  for (int y = y_begin; y < y_end; ++y) {
    for (int x = 0; x < out->width; ++x) {
      /* pixel centre, image-centred (src/reproject.cpp:287-288) */
      float cx = (x + 0.5f) - out->width * 0.5f;
      float cy = (y + 0.5f) - out->height * 0.5f;
      for (int c = 0; c < C; ++c) acc[c] = 0.0f;
      for (int ssx = 0; ssx < num_samples; ++ssx) {
        float scx = cx + (ssx + 1.0f) / (num_samples + 1.0f) - 0.5f; /* :295 */
        for (int ssy = 0; ssy < num_samples; ++ssy) {
          float scy = cy + (ssy + 1.0f) / (num_samples + 1.0f) - 0.5f; /* :298 */
          float sx, sy;
          source_position(in, out, rotation, scx, scy, &sx, &sy);
          if (interpolation == LRPO_NEAREST)
            tap_nearest(in, loop, sx, sy, tap);
          else if (interpolation == LRPO_BILINEAR)
            tap_bilinear(in, loop, sx, sy, tap);
          else
            tap_bicubic(in, loop, sx, sy, tap);
          for (int c = 0; c < C; ++c) acc[c] += tap[c]; /* :334-336 */
        }
        /* The reference stores after every ssx column (:338-341); only the last
         * store survives, and it is acc * normalize.  num_samples <= 0 therefore
         * leaves the output untouched, as here. */
        float *dst = out->data + y * pitch + (ptrdiff_t)x * C;
        for (int c = 0; c < C; ++c) dst[c] = acc[c] * normalize;
      }
    }
  }
  return LRPO_OK;
}

int lrpo_reproject(const lrpo_image *in, lrpo_image *out, int num_samples, int interpolation,
                   const float *rotation) {
  return lrpo_reproject_rows(in, out, num_samples, interpolation, rotation, 0, out->height);
}

int lrpo_source_coords(const lrpo_image *in, const lrpo_image *out, const float *rotation, float *sxy) {
  int rc = check_dispatch(in, out, LRPO_NEAREST);
  if (rc != LRPO_OK) return rc;
  for (int y = 0; y < out->height; ++y) {
    for (int x = 0; x < out->width; ++x) {
      float cx = (x + 0.5f) - out->width * 0.5f;
      float cy = (y + 0.5f) - out->height * 0.5f;
      float scx = cx + (0 + 1.0f) / (1 + 1.0f) - 0.5f;
      float scy = cy + (0 + 1.0f) / (1 + 1.0f) - 0.5f;
      float *d = sxy + ((ptrdiff_t)y * out->width + x) * 2;
      source_position(in, out, rotation, scx, scy, &d[0], &d[1]);
    }
  }
  return LRPO_OK;
}

/* post_process (src/reproject.cpp:421-437): exposure then extended Reinhard on
 * the first min(C,3) channels, in place. */
void lrpo_post_process(lrpo_image *img, float exposure, float reinhard) {
  int ch = img->channels < 3 ? img->channels : 3;
  ptrdiff_t n = (ptrdiff_t)img->width * img->height;
  float *p = img->data;
  for (ptrdiff_t i = 0; i < n; ++i, p += img->channels) {
    for (int c = 0; c < ch; ++c) {
      float v = p[c];
      v *= exposure;
      v = v * (1.0f + v / (reinhard * reinhard)) / (1.0f + v);
      p[c] = v;
    }
  }
}

/* computeRotationMatrix / multiplyMatrices (src/main.cpp:98-142): float sin/cos,
 * R = R_y(pan) * (R_x(pitch) * R_z(roll)), each product accumulated from 0 in
 * k order. */
static void mat3_mul(const float *a, const float *b, float *r) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0;
      for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
      r[i * 3 + j] = s;
    }
}
void lrpo_rotation_matrix(float pan, float pitch, float roll, float *out9) {
  float cxr = cosf(pitch), sxr = sinf(pitch);
  float cyr = cosf(pan), syr = sinf(pan);
  float czr = cosf(roll), szr = sinf(roll);
  float Rx[9] = {1, 0, 0, 0, cxr, -sxr, 0, sxr, cxr};
  float Ry[9] = {cyr, 0, syr, 0, 1, 0, -syr, 0, cyr};
  float Rz[9] = {czr, -szr, 0, szr, czr, 0, 0, 0, 1};
  float t[9];
  mat3_mul(Rx, Rz, t);
  mat3_mul(Ry, t, out9);
}

/* ---- synthetic frames (not in the reference) ---------------- */

static uint32_t mix32(uint32_t seed, uint32_t index) {
  uint32_t h = index * 0x9E3779B9u + seed;
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}

float lrpo_synth_value(uint32_t seed, uint32_t index, int kind) {
  uint32_t h = mix32(seed, index);
  float u = (float)(h >> 21) * (1.0f / 2048.0f); /* 11 significant bits: binary16-exact */
  if (kind == 0) return u;
  float d = 0.1f + u * 99.9f;
  /* round to 11 significant bits (what read_exr's HALF channels deliver) */
  union {
    float f;
    uint32_t u;
  } b;
  b.f = d;
  b.u = (b.u + 0x00000FFFu + ((b.u >> 13) & 1u)) & 0xFFFFE000u;
  return b.f;
}

void lrpo_synth_fill(float *data, int width, int height, int channels, uint32_t seed, int depth_channel) {
  uint32_t n = (uint32_t)width * (uint32_t)height;
  uint32_t idx = 0;
  for (uint32_t p = 0; p < n; ++p)
    for (int c = 0; c < channels; ++c, ++idx) data[idx] = lrpo_synth_value(seed, idx, c == depth_channel);
}

/* Host twin of lrp_checksum_device (csrc/lrp_aux_kernels.hip; not in the reference): the sum
 * mod 2^64 over all elements of a 64-bit hash of (bit pattern, index).  Order-independent. */
uint64_t lrpo_checksum(const float *data, size_t n) {
  const uint32_t *bits = (const uint32_t *)data;
  uint64_t acc = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t idx = (uint32_t)i;
    const uint32_t lo = mix32(bits[i], idx);
    const uint32_t hi = mix32(bits[i] ^ 0xA5A5A5A5u, idx * 2u + 0x7F4A7C15u);
    acc += ((uint64_t)hi << 32) | lo;
  }
  return acc;
}
