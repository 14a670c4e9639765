// lrp_source_axes.h — the horizontal and the vertical half of vec_to_rectilinear
// (src/reproject.cpp:160-167) and vec_to_equirectangular (:259-271), one function
// each, shared by the per-pixel code (lrp_kernel_v2.h) and by the per-column table
// builder (lrp_tables.hip) so that both execute the very same operations.
#pragma once

#include "lrp_math.h"

namespace lrp {

// cx (or cy) of vec_to_rectilinear for an already divided x / -z (or y / -z): :165-166
LRP_HD float rect_axis(float v, float extent, float sensor, float focal) { return v * extent / sensor * focal; }

// cx of vec_to_equirectangular: :262, :268
LRP_HD float equirect_cx(float x, float z, float lon_min, float lon_span, float img_w) {
  const float theta = -atan2f_(-x, -z);
  return ((theta - lon_min) / lon_span - 0.5f) * img_w;
}

// cy of vec_to_equirectangular: :263, :269
LRP_HD float equirect_phi(float x, float y, float z) { return asinf_(y / lrp_sqrtf(x * x + y * y + z * z)); }
LRP_HD float equirect_cy_of_phi(float phi, float lat_min, float lat_span, float img_h) {
  return ((phi - lat_min) / lat_span - 0.5f) * img_h;
}
LRP_HD float equirect_cy(float x, float y, float z, float lat_min, float lat_span, float img_h) {
  return equirect_cy_of_phi(equirect_phi(x, y, z), lat_min, lat_span, img_h);
}

// source texel coordinate from the lens-plane coordinate: :323-324
LRP_HD float texel_coord(float c, float extent) { return (c - 0.5f) + extent * 0.5f; }

} // namespace lrp
