// lrp_host_util.cpp — host-side producers of hot-path inputs, restating the
// reference CLI helpers (src/main.cpp:15-142).  No device code.
#include <cmath>
#include <cstring>

#include "../../include/lrp.h"

namespace {
// multiplyMatrices, src/main.cpp:98-107: accumulate from 0 in k order.
void mat3_mul(const float *a, const float *b, float *r) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0;
      for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
      r[i * 3 + j] = s;
    }
}
} // namespace

extern "C" {

void lrp_rotation_matrix(float pan, float pitch, float roll, float *out9) {
  // computeRotationMatrix, src/main.cpp:110-142 (rot_x = pitch, rot_y = pan, rot_z = roll)
  const float cx = std::cos(pitch), sx = std::sin(pitch);
  const float cy = std::cos(pan), sy = std::sin(pan);
  const float cz = std::cos(roll), sz = std::sin(roll);
  const float Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
  const float Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
  const float Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  float t[9];
  mat3_mul(Rx, Rz, t);
  mat3_mul(Ry, t, out9);
}

void lrp_lens_rectilinear(lrp_lens *lens, float focal_length, float sensor_width, float res_x, float res_y) {
  std::memset(lens, 0, sizeof(*lens));
  lens->type = LRP_RECTILINEAR;
  lens->u.rectilinear.focal_length = focal_length;
  lens->sensor_width = sensor_width;
  lens->sensor_height = res_y / res_x * sensor_width; // src/main.cpp:27
}

void lrp_lens_equidistant(lrp_lens *lens, float fov) {
  std::memset(lens, 0, sizeof(*lens));
  lens->type = LRP_FISHEYE_EQUIDISTANT;
  lens->u.fisheye_equidistant.fov = fov;
  lens->sensor_width = 36.0f; // src/main.cpp:53-54
  lens->sensor_height = 36.0f;
}

void lrp_lens_equirectangular(lrp_lens *lens, float longitude_min, float longitude_max, float latitude_min,
                              float latitude_max) {
  std::memset(lens, 0, sizeof(*lens));
  lens->type = LRP_EQUIRECTANGULAR;
  lens->u.equirectangular.longitude_min = longitude_min;
  lens->u.equirectangular.longitude_max = longitude_max;
  lens->u.equirectangular.latitude_min = latitude_min;
  lens->u.equirectangular.latitude_max = latitude_max;
  lens->sensor_width = lens->sensor_height = 0; // src/main.cpp:93
}

void lrp_lens_equirectangular_full(lrp_lens *lens) {
  // src/main.cpp:62-66: double expressions narrowed to float on assignment
  lrp_lens_equirectangular(lens, (float)(-M_PI), (float)(M_PI), (float)(-M_PI * 0.5f), (float)(M_PI * 0.5f));
}

} // extern "C"
