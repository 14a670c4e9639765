// lrp_config.h — lens configuration <-> Blender-style JSON (row f4; reference
// src/config.cpp:7-106, templates in the reference README.md:159-225).
#pragma once
#include "lrp.h"
#include "lrp_json.h"

namespace lrp_cfg {
// extract_lens_info_from_config: "camera", "sensor_size" (and "resolution") of the root object.
// Throws std::invalid_argument("Unknown camera_type") / ("Unknown lens_unit") like the reference.
lrp_lens extract_lens_info_from_config(const lrp_json::Value &cfg);
// store_lens_info_in_config: rewrites "camera" and "sensor_size"; every other key is kept.
void store_lens_info_in_config(const lrp_lens &lens, lrp_json::Value &out_cfg);
} // namespace lrp_cfg
