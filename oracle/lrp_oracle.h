/*
 * lrp_oracle.h — CPU restatement of the reference's per-pixel lens
 * reprojection path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * Nothing under image-lens-reproject_amd/ or include/ links, loads or calls it.
 *
 * PINNING STATUS: **parity unpinned**.
 *   - The reference ships no tests, golden vectors or fixtures for this path
 *     (SURVEY.md §4, §8c).
 *   - The reference translation unit (src/reproject.cpp) cannot be compiled in
 *     this image: it includes <tracy/Tracy.hpp> (src/reproject.cpp:6) and,
 *     through src/config.hpp:3, <nlohmann/json_fwd.hpp>; both come from git
 *     submodules that are empty in the checkout (lib/tracy, lib/json) and are
 *     installed nowhere on the image.  Writing stand-in headers is not an
 *     acceptable way to make a reference build, so there is no oracle/_ref.
 *   What IS pinned: every libm routine the path depends on is reproduced bit
 *   for bit and checked exhaustively against the live glibc 2.35 of this image
 *   (tests/test_math_vs_libm.py), and this file follows the reference source
 *   line by line with the file:line of every step cited in lrp_oracle.c.
 *
 * The arithmetic is IEEE binary32, un-fused, 32-bit int indexing, exactly as a
 * `g++ -O3` (no -march, no fast-math) build of the reference executes on
 * x86-64; transcendental functions are the host libm's (sinf, cosf -> merged
 * into sincosf by GCC, atanf, atan2f, asinf, sqrtf), as in the reference.
 */
#ifndef LRP_ORACLE_H
#define LRP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same numbering as reference src/config.hpp:7-13. */
enum {
  LRPO_RECTILINEAR = 0,
  LRPO_FISHEYE_EQUIDISTANT = 1,
  LRPO_FISHEYE_EQUISOLID = 2,
  LRPO_FISHEYE_STEREOGRAPHIC = 3,
  LRPO_EQUIRECTANGULAR = 4
};
/* Same numbering as reference src/reproject.hpp:16-20. */
enum { LRPO_NEAREST = 0, LRPO_BILINEAR = 1, LRPO_BICUBIC = 2 };

/* Layout-identical to reference LensInfo (src/config.hpp:15-37): type @0,
 * 16-byte union @4, sensor_width @20, sensor_height @24, sizeof == 28.
 *   rectilinear:      u[0] = focal_length
 *   equidistant:      u[0] = fov
 *   equisolid:        u[0] = focal_length, u[1] = fov
 *   equirectangular:  u[0..3] = latitude_min, latitude_max, longitude_min, longitude_max */
typedef struct lrpo_lens {
  int32_t type;
  float u[4];
  float sensor_width;
  float sensor_height;
} lrpo_lens;

/* Layout-identical to reference Image (src/reproject.hpp:9-14), sizeof == 56. */
typedef struct lrpo_image {
  lrpo_lens lens;
  int32_t width, height, channels;
  float *data; /* interleaved, row-major, data[(y*W + x)*C + c] */
  int32_t data_layout;
} lrpo_image;

enum {
  LRPO_OK = 0,
  LRPO_ERR_OUTPUT_LENS = 1,  /* reference: "Output lens type not supported." + exit(1) */
  LRPO_ERR_INPUT_LENS = 2,   /* reference: "Input lens type not supported." + exit(1) */
  LRPO_ERR_INTERPOLATION = 3 /* reference: "Interpolation method not supported." + exit(1) */
};

/* reproject() over output rows [y_begin, y_end) — rows are independent in the
 * reference loop (src/reproject.cpp:284), so a band computes the same bits as
 * the full call.  rotation may be NULL (src/reproject.cpp:303). */
int lrpo_reproject_rows(const lrpo_image *in, lrpo_image *out, int num_samples, int interpolation,
                        const float *rotation, int y_begin, int y_end);

/* Whole image == reference reproject() (src/reproject.cpp:405-419). */
int lrpo_reproject(const lrpo_image *in, lrpo_image *out, int num_samples, int interpolation,
                   const float *rotation);

/* reference post_process() (src/reproject.cpp:421-437). */
void lrpo_post_process(lrpo_image *img, float exposure, float reinhard);

/* Source coordinates only (diagnostics): for every output pixel and sub-sample
 * 0 writes the top-left-origin (sx, sy) the sampler would receive. */
int lrpo_source_coords(const lrpo_image *in, const lrpo_image *out, const float *rotation,
                       float *sxy /* outH*outW*2 */);

/* Caller-side helpers restated from the reference CLI (src/main.cpp:98-142):
 * R = R_y(pan) * R_x(pitch) * R_z(roll), row-major, angles in radians. */
void lrpo_rotation_matrix(float pan, float pitch, float roll, float *out9);

/* Counter-based synthetic frames: value of float element
 * `index` of a frame with seed `seed`.  kind 0 = colour in [0,1) with 11
 * significant bits, kind 1 = depth in [0.1,100) rounded to binary16 precision. */
float lrpo_synth_value(uint32_t seed, uint32_t index, int kind);
void lrpo_synth_fill(float *data, int width, int height, int channels, uint32_t seed,
                     int depth_channel /* -1 = none */);
/* Host twin of the product's lrp_checksum_device (include/lrp.h): order-independent 64-bit
 * checksum of the bit patterns of n floats (fixtures and bench checks; not in the reference). */
uint64_t lrpo_checksum(const float *data, size_t n);

#ifdef __cplusplus
}
#endif
#endif
