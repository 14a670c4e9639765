// lrp_run_plan.cpp — see lrp_run_plan.h.  Messages and exit codes are the reference's
// (src/main.cpp:256-535); the lens flags of both sides go through one table.
#include "lrp_run_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "lrp_config.h"

namespace fs = std::filesystem;

namespace lrp_cli {

namespace {

std::vector<std::string> split(const std::string &text, char sep) {
  std::vector<std::string> parts;
  size_t begin = 0;
  for (;;) {
    const size_t at = text.find(sep, begin);
    parts.push_back(text.substr(begin, at == std::string::npos ? std::string::npos : at - begin));
    if (at == std::string::npos) return parts;
    begin = at + 1;
  }
}
float number(const std::string &s) { return (float)std::atof(s.c_str()); } // parsed as double, narrowed: src/main.cpp:15-95

// ---- lens flags --------------------------------------------------------------------------------
// One description per lens model; the input side prefixes the flag with "i-".  `parse` gets the
// flag's value split at commas and the resolution the lens is for.
struct LensFlag {
  const char *flag;
  bool (*parse)(const std::string &text, float res_x, float res_y, lrp_lens &lens);
};

bool lens_rectilinear(const std::string &text, float res_x, float res_y, lrp_lens &lens) {
  const std::vector<std::string> v = split(text, ',');
  if (v.size() < 2) {
    std::printf("Error: Required format for --rectilinear focal_len,sensor_width\n");
    return false;
  }
  // the second value runs to the end of the text (src/main.cpp:24-25)
  lrp_lens_rectilinear(&lens, number(v[0]), number(text.substr(v[0].size() + 1)), res_x, res_y);
  return true;
}

bool lens_equisolid(const std::string &text, float res_x, float res_y, lrp_lens &lens) {
  const size_t c1 = text.find(','), c2 = c1 == std::string::npos ? std::string::npos : text.find(',', c1 + 1);
  if (c2 == std::string::npos) {
    std::printf("Error: Required format for --equisolid focal_len,sensor_width,fov\n");
    return false;
  }
  std::memset(&lens, 0, sizeof(lens));
  lens.type = LRP_FISHEYE_EQUISOLID;
  lens.u.fisheye_equisolid.focal_length = number(text.substr(0, c1));
  lens.u.fisheye_equisolid.fov = number(text.substr(c2 + 1));
  lens.sensor_width = number(text.substr(c1 + 1, c2)); // (sic) length = c2, src/main.cpp:45: atof stops at the comma anyway
  lens.sensor_height = res_y / res_x * lens.sensor_width;
  return true;
}

bool lens_equidistant(const std::string &text, float, float, lrp_lens &lens) {
  lrp_lens_equidistant(&lens, number(text));
  return true;
}

// "full" or longitude_min,longitude_max,latitude_min,latitude_max in radians (src/main.cpp:58-95)
bool lens_equirectangular(const std::string &text, float, float, lrp_lens &lens) {
  if (text == "full") {
    lrp_lens_equirectangular_full(&lens);
    return true;
  }
  const std::vector<std::string> v = split(text, ',');
  if (v.size() != 4) {
    std::printf("Error: expected 4 arguments for equirectangular, got %d.\n", (int)v.size());
    return false;
  }
  lrp_lens_equirectangular(&lens, number(v[0]), number(v[1]), number(v[2]), number(v[3]));
  return true;
}

const LensFlag kLensFlags[] = {{"rectilinear", lens_rectilinear},
                               {"equisolid", lens_equisolid},
                               {"equidistant", lens_equidistant},
                               {"equirectangular", lens_equirectangular}};

// Applies every lens flag that was given (in table order, the last one wins like in the reference) and
// returns how many there were; -1 after a format error.
int apply_lens_flags(const CommandLine &cl, const char *prefix, float res_x, float res_y, lrp_lens &lens) {
  int found = 0;
  for (const LensFlag &lf : kLensFlags) {
    const std::string flag = std::string(prefix) + lf.flag;
    if (!cl.has(flag)) continue;
    if (!lf.parse(cl[flag], res_x, res_y, lens)) return -1;
    ++found;
  }
  return found;
}

// ---- resolvers ---------------------------------------------------------------------------------
int usage_error(const std::string &message, const char *argv0) {
  std::printf("%s\n\n%s\n", message.c_str(), help_text(argv0).c_str());
  return 1;
}

int resolve_io(const CommandLine &cl, const char *argv0, RunPlan &p) {
  if (cl.has("input-dir") && cl.has("single")) {
    std::printf("Error: cannot specify both --input-dir and --single.\n%s\n", help_text(argv0).c_str());
    return 1;
  }
  if (cl.has("input-dir"))
    p.input_dir = cl["input-dir"];
  else if (cl.has("single"))
    p.input_single = cl["single"];
  else {
    std::printf("Error: No input specified.\n");
    return 1;
  }
  if (!cl.has("output-dir")) return usage_error("Option 'output-dir' has no value", argv0);
  p.output_dir = cl["output-dir"];
  p.filter_prefix = cl["filter-prefix"];
  p.filter_suffix = cl["filter-suffix"];
  p.write_exr = cl.has("exr");
  p.write_png = cl.has("png");
  p.skip_if_exists = cl.has("skip-if-exists");
  p.dry_run = cl.has("dry-run");
  p.workers = std::max(1, std::atoi(cl["parallel"].c_str()));
  p.first_device = std::atoi(cl["device"].c_str());
  p.gpus = std::max(1, std::atoi(cl["gpus"].c_str()));
  p.streams = std::max(0, std::min(64, std::atoi(cl["streams"].c_str())));
  return 0;
}

int resolve_rendering(const CommandLine &cl, const char *, RunPlan &p) {
  p.num_samples = std::atoi(cl["samples"].c_str());
  if (cl.has("output-resolution")) {
    const std::string arg = cl["output-resolution"];
    const size_t comma = arg.find(',');
    if (comma == std::string::npos || comma == arg.length() - 1 || comma == 0) {
      std::printf("Error: Specify both width and height, separated by a comma in output-resolution.\n");
      return 1;
    }
    p.out_width = std::atoi(arg.substr(0, comma).c_str());
    p.out_height = std::atoi(arg.substr(comma + 1).c_str());
  } else {
    p.scale = std::atof(cl["scale"].c_str());
  }
  // --rotation pan,pitch,roll: degrees -> radians in double, narrowed to float; missing angles read as
  // whatever atof makes of the text the reference slices out (src/main.cpp:312-325)
  const std::string euler = cl["rotation"];
  const size_t c0 = euler.find(','), c1 = euler.find(',', c0 + 1);
  const double deg[3] = {std::atof(euler.substr(0, c0).c_str()), std::atof(euler.substr(c0 + 1, c1).c_str()),
                         std::atof(euler.substr(c1 + 1).c_str())};
  lrp_rotation_matrix((float)(deg[0] / 180.0 * M_PI), (float)(deg[1] / 180.0 * M_PI), (float)(deg[2] / 180.0 * M_PI), p.rotation);
  p.exposure = std::pow(2.0, std::atof(cl["exposure"].c_str()));
  p.reinhard = std::atof(cl["reinhard"].c_str());
  p.reproject = !cl.has("no-reproject");
  return 0;
}

int resolve_interpolation(const CommandLine &cl, const char *argv0, RunPlan &p) {
  // several flags only warn, the last in the order nn, bl, bc wins (src/main.cpp:359-376)
  static const struct { const char *flag; int value; } kInterp[] = {{"nn", LRP_NEAREST}, {"bl", LRP_BILINEAR}, {"bc", LRP_BICUBIC}};
  int given = 0;
  for (const auto &i : kInterp)
    if (cl.has(i.flag)) {
      p.interpolation = i.value;
      ++given;
    }
  if (given > 1) std::printf("Cannot specify more than one interpolation method.\n\n%s", help_text(argv0).c_str());
  return 0;
}

int resolve_input_lens(const CommandLine &cl, const char *argv0, RunPlan &p) {
  if (cl.has("no-configs")) {
    const std::vector<std::string> wh = split(cl["no-configs"], ',');
    p.in_width = std::atoi(wh[0].c_str());
    // without a comma the reference's find() yields -1 and substr(comma + 1) is the whole string again: height = width
    // (src/main.cpp:389-391)
    p.in_height = std::atoi(wh.size() > 1 ? cl["no-configs"].substr(wh[0].size() + 1).c_str() : cl["no-configs"].c_str());
    const int found = apply_lens_flags(cl, "i-", (float)p.in_width, (float)p.in_height, p.input_lens);
    if (found < 0) return 1;
    if (found > 1) {
      std::printf("Error: only specify one input lens type: [--i-rectilinear, --i-equisolid, --i-equidistant, "
                  "--i-equirectangular].\n");
      return 1;
    }
    return 0;
  }
  // config-file mode, src/main.cpp:429-443
  if (!cl.has("input-cfg") || !cl.has("output-cfg")) {
    std::printf("Option 'input-cfg' / 'output-cfg' has no value (or use --no-configs width,height)\n\n%s\n", help_text(argv0).c_str());
    return 1;
  }
  p.has_config = true;
  p.output_config_path = cl["output-cfg"];
  try {
    std::ifstream in(cl["input-cfg"]);
    if (!in) throw std::invalid_argument("cannot open " + cl["input-cfg"]);
    std::stringstream text;
    text << in.rdbuf();
    p.config = lrp_json::parse(text.str());
    std::printf("Found camera config: %s\n", p.config.at("camera").dump(1).c_str());
    p.in_width = p.config.at("resolution").at(0).as_int();
    p.in_height = p.config.at("resolution").at(1).as_int();
    p.input_lens = lrp_cfg::extract_lens_info_from_config(p.config);
  } catch (const std::exception &e) { // the reference lets the exception end the program
    std::printf("Error: %s\n", e.what());
    return 1;
  }
  return 0;
}

int resolve_output_lens(const CommandLine &cl, RunPlan &p) {
  if (p.out_width == 0 && p.out_height == 0) { // src/main.cpp:448-451: rounded towards zero
    p.out_width = int(p.in_width * p.scale);
    p.out_height = int(p.in_height * p.scale);
  }
  int found = apply_lens_flags(cl, "", (float)p.out_width, (float)p.out_height, p.output_lens);
  if (found < 0) return 1;
  if (!p.reproject) {
    p.output_lens = p.input_lens;
    ++found;
  }
  if (found > 1) {
    std::printf("Error: only specify one output lens type: [--rectilinear, --equisolid, --equidistant, "
                "--equirectangular, --no-reproject].\n");
    return 1;
  }
  return 0;
}

} // namespace

int resolve_run_plan(const CommandLine &cl, const char *argv0, RunPlan &plan) {
  if (int rc = resolve_io(cl, argv0, plan)) return rc;
  if (int rc = resolve_rendering(cl, argv0, plan)) return rc;
  if (!plan.write_exr && !plan.write_png) {
    std::printf("Error: Did not specify any output format.\nChoose --png or --exr. (both are possible).\n");
    return 1;
  }
  if (int rc = resolve_interpolation(cl, argv0, plan)) return rc;
  if (int rc = resolve_input_lens(cl, argv0, plan)) return rc;
  return resolve_output_lens(cl, plan);
}

bool name_passes_filters(const RunPlan &p, const std::string &name) {
  if (name.size() < p.filter_prefix.size() || name.size() < p.filter_suffix.size()) return false;
  return name.compare(0, p.filter_prefix.size(), p.filter_prefix) == 0 &&
         name.compare(name.size() - p.filter_suffix.size(), p.filter_suffix.size(), p.filter_suffix) == 0;
}

int write_output_config(RunPlan &plan) {
  try {
    lrp_cfg::store_lens_info_in_config(plan.output_lens, plan.config);
    plan.config["resolution"][0] = lrp_json::Value::integer(plan.out_width);
    plan.config["resolution"][1] = lrp_json::Value::integer(plan.out_height);
    if (plan.config.contains("frames")) {
      auto &frames = plan.config["frames"].arr;
      frames.erase(std::remove_if(frames.begin(), frames.end(),
                                  [&](const lrp_json::Value &f) { return !name_passes_filters(plan, f.at("name").str()); }),
                   frames.end());
    }
    std::printf("Saving output config: %s\n", plan.output_config_path.c_str());
    std::ofstream out(plan.output_config_path);
    out << plan.config.dump(2);
  } catch (const std::exception &e) {
    std::printf("Error: %s\n", e.what());
    return 1;
  }
  return 0;
}

std::vector<fs::path> enumerate_inputs(const RunPlan &plan) {
  std::vector<fs::path> files;
  if (plan.input_dir.empty()) {
    files.push_back(fs::path(plan.input_single));
    return files;
  }
  for (const auto &entry : fs::directory_iterator(fs::path(plan.input_dir))) {
    const fs::path &p = entry.path();
    if (!entry.is_regular_file() || !name_passes_filters(plan, p.filename().string())) continue;
    if (p.extension() == ".exr" || p.extension() == ".png") files.push_back(p); // src/main.cpp:645-649
  }
  std::sort(files.begin(), files.end());
  return files;
}

} // namespace lrp_cli
