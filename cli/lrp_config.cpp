// lrp_config.cpp — see lrp_config.h.  The Blender-style camera description of reference
// src/config.cpp:7-106 as ONE schema table (camera type / panorama type <-> lens model, JSON key <->
// lens parameter) that both directions walk.  The quirks of the reference live in the table:
// the equisolid focal length key is "fisheye_lens" (the reference README says "lens"); an
// equirectangular lens is WRITTEN with "panorama_type": "RECTILINEAR" (src/config.cpp:98) but only
// READ as "EQUIRECTANGULAR" (:29); a PANO camera of any other panorama_type leaves the lens type
// unset in the reference — rejected here.
#include "lrp_config.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace lrp_cfg {

namespace {

struct Param {
  const char *key; // in "camera"
  int slot;        // float index into lrp_lens::u (the union payload, include/lrp.h)
};

struct Model {
  int lens_type;
  const char *camera_type;  // "camera"."type"
  const char *pano_read;    // "camera"."panorama_type" accepted when reading (nullptr: not a panorama)
  const char *pano_written; // ... and the one written
  Param params[4];
  int n_params;
};

// u.rectilinear {focal_length}; u.fisheye_equidistant {fov}; u.fisheye_equisolid {focal_length, fov};
// u.equirectangular {latitude_min, latitude_max, longitude_min, longitude_max}
const Model kModels[] = {
    {LRP_RECTILINEAR, "PERSP", nullptr, nullptr, {{"focal_length", 0}}, 1},
    {LRP_FISHEYE_EQUIDISTANT, "PANO", "FISHEYE_EQUIDISTANT", "FISHEYE_EQUIDISTANT", {{"fisheye_fov", 0}}, 1},
    {LRP_FISHEYE_EQUISOLID, "PANO", "FISHEYE_EQUISOLID", "FISHEYE_EQUISOLID", {{"fisheye_lens", 0}, {"fisheye_fov", 1}}, 2},
    {LRP_EQUIRECTANGULAR, "PANO", "EQUIRECTANGULAR", "RECTILINEAR" /* (sic) */,
     {{"latitude_min", 0}, {"latitude_max", 1}, {"longitude_min", 2}, {"longitude_max", 3}}, 4},
};

static_assert(sizeof(((lrp_lens *)nullptr)->u) == 4 * sizeof(float), "lens payload: four floats");

// The perspective camera may give its lens as an angle instead of a focal length (src/config.cpp:41-52).
void read_perspective_focal(const lrp_json::Value &cam, lrp_lens &lens) {
  const std::string unit = cam.at("lens_unit").str();
  if (unit == "MILLIMETERS") {
    lens.u.raw[0] = cam.at("focal_length").as_float();
  } else if (unit == "FOV") {
    const float angle = cam.at("angle").as_float();
    std::printf("Warning: relying on 'angle' is unsafe. Angle is assumed to be based on the width of the sensor.\n");
    lens.u.raw[0] = lens.sensor_width / std::tan(0.5f * angle);
  } else {
    throw std::invalid_argument("Unknown lens_unit");
  }
}

// ... and is written with a synthetic OpenGL-style projection matrix whose clip planes are invented
// (src/config.cpp:69-83).
lrp_json::Value projection_matrix(const lrp_lens &lens) {
  using lrp_json::Value;
  const float focal = lens.u.raw[0], near_plane = 0.1f, far_plane = 100.0f;
  float m[4][4] = {};
  m[0][0] = 2.0f * focal / lens.sensor_width;
  m[1][1] = 2.0f * focal / lens.sensor_height;
  m[2][2] = -(far_plane + near_plane) / (far_plane - near_plane);
  m[2][3] = -2.0f * far_plane * near_plane / (far_plane - near_plane);
  m[3][2] = -1.0f;
  Value rows = Value::array();
  for (const auto &row : m) {
    Value r = Value::array();
    for (float v : row) r.arr.push_back(Value::real(v));
    rows.arr.push_back(r);
  }
  return rows;
}

} // namespace

lrp_lens extract_lens_info_from_config(const lrp_json::Value &cfg) {
  const lrp_json::Value &cam = cfg.at("camera");
  lrp_lens lens;
  std::memset(&lens, 0, sizeof(lens));
  lens.sensor_width = cfg.at("sensor_size").at(0).as_float();
  lens.sensor_height = cfg.at("sensor_size").at(1).as_float();
  for (int axis = 0; axis < 2; ++axis) (void)cfg.at("resolution").at(axis).as_int(); // required, unused (src/config.cpp:15-16)
  const std::string camera_type = cam.at("type").str();
  const bool pano = camera_type == "PANO";
  const std::string pano_type = pano ? cam.at("panorama_type").str() : std::string();
  for (const Model &m : kModels) {
    if (camera_type != m.camera_type || (pano && pano_type != m.pano_read)) continue;
    lens.type = m.lens_type;
    if (m.lens_type == LRP_RECTILINEAR)
      read_perspective_focal(cam, lens);
    else
      for (int i = 0; i < m.n_params; ++i) lens.u.raw[m.params[i].slot] = cam.at(m.params[i].key).as_float();
    return lens;
  }
  throw std::invalid_argument(pano ? "Unknown panorama_type" : "Unknown camera_type");
}

void store_lens_info_in_config(const lrp_lens &lens, lrp_json::Value &out_cfg) {
  using lrp_json::Value;
  for (const Model &m : kModels) {
    if (m.lens_type != lens.type) continue;
    out_cfg["camera"] = Value::object();
    out_cfg["sensor_size"][0] = Value::real(lens.sensor_width);
    out_cfg["sensor_size"][1] = Value::real(lens.sensor_height);
    Value &cam = out_cfg["camera"];
    cam["type"] = Value::string(m.camera_type);
    if (m.pano_written) cam["panorama_type"] = Value::string(m.pano_written);
    for (int i = 0; i < m.n_params; ++i) cam[m.params[i].key] = Value::real(lens.u.raw[m.params[i].slot]);
    if (m.lens_type == LRP_RECTILINEAR) {
      cam["lens_unit"] = Value::string("MILLIMETERS");
      cam["projection_matrix"] = projection_matrix(lens);
    }
    return;
  }
  throw std::invalid_argument("Unsupported lens type.");
}

} // namespace lrp_cfg
