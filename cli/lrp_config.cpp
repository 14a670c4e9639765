// lrp_config.cpp — see lrp_config.h.  Key names and quirks follow the reference:
// the equisolid focal length key is "fisheye_lens" (the reference README says "lens"), an
// equirectangular lens is written with "panorama_type": "RECTILINEAR" (src/config.cpp:98) and
// read back only as "EQUIRECTANGULAR" (:29); a PANO camera of any other panorama_type leaves
// the lens type unset in the reference — rejected here.
#include "lrp_config.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace lrp_cfg {

lrp_lens extract_lens_info_from_config(const lrp_json::Value &cfg) {
  const lrp_json::Value &cam = cfg.at("camera");
  std::string camera_type = cam.at("type").str();
  lrp_lens lens;
  std::memset(&lens, 0, sizeof(lens));
  lens.sensor_width = cfg.at("sensor_size").at(0).as_float();
  lens.sensor_height = cfg.at("sensor_size").at(1).as_float();
  (void)cfg.at("resolution").at(0).as_int(); // read (and required) by the reference, src/config.cpp:15-16
  (void)cfg.at("resolution").at(1).as_int();
  if (camera_type == "PANO") {
    camera_type = cam.at("panorama_type").str();
    if (camera_type == "FISHEYE_EQUIDISTANT") {
      lens.type = LRP_FISHEYE_EQUIDISTANT;
      lens.u.fisheye_equidistant.fov = cam.at("fisheye_fov").as_float();
    } else if (camera_type == "FISHEYE_EQUISOLID") {
      lens.type = LRP_FISHEYE_EQUISOLID;
      lens.u.fisheye_equisolid.focal_length = cam.at("fisheye_lens").as_float();
      lens.u.fisheye_equisolid.fov = cam.at("fisheye_fov").as_float();
    } else if (camera_type == "EQUIRECTANGULAR") {
      lens.type = LRP_EQUIRECTANGULAR;
      lens.u.equirectangular.latitude_min = cam.at("latitude_min").as_float();
      lens.u.equirectangular.latitude_max = cam.at("latitude_max").as_float();
      lens.u.equirectangular.longitude_min = cam.at("longitude_min").as_float();
      lens.u.equirectangular.longitude_max = cam.at("longitude_max").as_float();
    } else {
      throw std::invalid_argument("Unknown panorama_type");
    }
  } else if (camera_type == "PERSP") {
    lens.type = LRP_RECTILINEAR;
    const std::string lens_unit = cam.at("lens_unit").str();
    if (lens_unit == "MILLIMETERS") {
      lens.u.rectilinear.focal_length = cam.at("focal_length").as_float();
    } else if (lens_unit == "FOV") {
      const float angle = cam.at("angle").as_float();
      std::printf("Warning: relying on 'angle' is unsafe. Angle is assumed to be based on the width of the sensor.\n");
      lens.u.rectilinear.focal_length = lens.sensor_width / std::tan(0.5f * angle); // src/config.cpp:47-48
    } else {
      throw std::invalid_argument("Unknown lens_unit");
    }
  } else {
    throw std::invalid_argument("Unknown camera_type");
  }
  return lens;
}

void store_lens_info_in_config(const lrp_lens &ol, lrp_json::Value &out_cfg) {
  using lrp_json::Value;
  out_cfg["camera"] = Value::object();
  out_cfg["sensor_size"][0] = Value::real(ol.sensor_width);
  out_cfg["sensor_size"][1] = Value::real(ol.sensor_height);
  Value &cam = out_cfg["camera"];
  if (ol.type == LRP_RECTILINEAR) {
    const float focal = ol.u.rectilinear.focal_length;
    cam["type"] = Value::string("PERSP");
    cam["lens_unit"] = Value::string("MILLIMETERS");
    cam["focal_length"] = Value::real(focal);
    // a synthetic OpenGL-style projection matrix with invented clip planes (src/config.cpp:69-83)
    float proj[16] = {0.0f};
    proj[0] = 2.0f * focal / ol.sensor_width;
    proj[5] = 2.0f * focal / ol.sensor_height;
    proj[14] = -1.0f;
    const float near_plane = 0.1f, far_plane = 100.0f;
    proj[10] = -(far_plane + near_plane) / (far_plane - near_plane);
    proj[11] = -2.0f * far_plane * near_plane / (far_plane - near_plane);
    Value m = Value::array();
    for (int r = 0; r < 4; ++r) {
      Value row = Value::array();
      for (int c = 0; c < 4; ++c) row.arr.push_back(Value::real(proj[r * 4 + c]));
      m.arr.push_back(row);
    }
    cam["projection_matrix"] = m;
  } else if (ol.type == LRP_FISHEYE_EQUISOLID) {
    cam["type"] = Value::string("PANO");
    cam["panorama_type"] = Value::string("FISHEYE_EQUISOLID");
    cam["fisheye_lens"] = Value::real(ol.u.fisheye_equisolid.focal_length);
    cam["fisheye_fov"] = Value::real(ol.u.fisheye_equisolid.fov);
  } else if (ol.type == LRP_FISHEYE_EQUIDISTANT) {
    cam["type"] = Value::string("PANO");
    cam["panorama_type"] = Value::string("FISHEYE_EQUIDISTANT");
    cam["fisheye_fov"] = Value::real(ol.u.fisheye_equidistant.fov);
  } else if (ol.type == LRP_EQUIRECTANGULAR) {
    cam["type"] = Value::string("PANO");
    cam["panorama_type"] = Value::string("RECTILINEAR"); // (sic) src/config.cpp:98
    cam["latitude_min"] = Value::real(ol.u.equirectangular.latitude_min);
    cam["latitude_max"] = Value::real(ol.u.equirectangular.latitude_max);
    cam["longitude_min"] = Value::real(ol.u.equirectangular.longitude_min);
    cam["longitude_max"] = Value::real(ol.u.equirectangular.longitude_max);
  } else {
    throw std::invalid_argument("Unsupported lens type.");
  }
}

} // namespace lrp_cfg
