// lrp_cli_options.cpp — see lrp_cli_options.h.
#include "lrp_cli_options.h"

#include <cstring>
#include <stdexcept>

namespace lrp_cli {

namespace {

struct Option {
  const char *name;  // long name
  char short_name;   // 0 = none
  bool takes_value;
  const char *default_value; // nullptr = none
  const char *arg_help, *help, *group;
};

const Option kOptions[] = {
    {"input-cfg", 0, true, nullptr, "json-file", "Input JSON file with the lens and camera settings of the input images.", "Input/output"},
    {"output-cfg", 0, true, nullptr, "json-file", "Output JSON file: the input config with the output lens and resolution.", "Input/output"},
    {"no-configs", 0, true, nullptr, "width,height", "Work without config files: input lens from the --i-* flags, input resolution given here.", "Input/output"},
    {"input-dir", 'i', true, nullptr, "dir", "Directory with the images to reproject (.exr, .png; --single also takes .jpg / .jpeg).", "Input/output"},
    {"single", 0, true, nullptr, "file", "A single input file to convert.", "Input/output"},
    {"output-dir", 'o', true, nullptr, "dir", "Directory for the reprojected images.", "Input/output"},
    {"exr", 0, false, nullptr, "", "Write EXR files (colour and depth).", "Input/output"},
    {"png", 0, false, nullptr, "", "Write PNG files (colour only).", "Input/output"},
    {"filter-prefix", 0, true, "", "prefix", "Only files whose name starts with this.", "Filter files"},
    {"filter-suffix", 0, true, "", "suffix", "Only files whose name ends with this.", "Filter files"},
    {"samples", 's', true, "1", "number", "Sub-samples per dimension and output pixel.", "Sampling"},
    {"nn", 0, false, nullptr, "", "Nearest-neighbour interpolation.", "Sampling"},
    {"bl", 0, false, nullptr, "", "Bilinear interpolation.", "Sampling"},
    {"bc", 0, false, nullptr, "", "Bicubic interpolation (default).", "Sampling"},
    {"scale", 0, true, "1.0", "fraction", "Output size as a fraction of the input size (rounded towards zero); raise --samples when down-scaling.", "Sampling"},
    {"output-resolution", 0, true, nullptr, "width,height", "Fixed output resolution; overrides --scale.", "Sampling"},
    {"i-rectilinear", 0, true, nullptr, "focal_length,sensor_width", "Input images are rectilinear.", "Input optics (with --no-configs)"},
    {"i-equisolid", 0, true, nullptr, "focal_length,sensor_width,fov", "Input images are equisolid fisheye (parsed; reproject() rejects it, as in the reference).", "Input optics (with --no-configs)"},
    {"i-equidistant", 0, true, nullptr, "fov", "Input images are equidistant fisheye.", "Input optics (with --no-configs)"},
    {"i-equirectangular", 0, true, nullptr, "long_min,long_max,lat_min,lat_max | full", "Input images are equirectangular (radians).", "Input optics (with --no-configs)"},
    {"no-reproject", 0, false, nullptr, "", "Keep the input lens (scaling / colour processing only).", "Output optics"},
    {"rectilinear", 0, true, nullptr, "focal_length,sensor_width", "Output rectilinear images.", "Output optics"},
    {"equisolid", 0, true, nullptr, "focal_length,sensor_width,fov", "Output equisolid images (rejected by reproject(), as in the reference).", "Output optics"},
    {"equidistant", 0, true, nullptr, "fov", "Output equidistant fisheye images.", "Output optics"},
    {"equirectangular", 0, true, nullptr, "long_min,long_max,lat_min,lat_max | full", "Output equirectangular images.", "Output optics"},
    {"rotation", 0, true, "0.0", "pan,pitch,roll (degrees)", "Rotate the view.", "Output optics"},
    {"exposure", 0, true, "0.0", "EV", "Exposure compensation in stops.", "Color processing"},
    {"reinhard", 0, true, "1.0", "max", "Reinhard tone mapping with this maximum (after exposure).", "Color processing"},
    {"skip-if-exists", 0, false, nullptr, "", "Skip files whose outputs already exist.", "Runtime"},
    {"parallel", 'j', true, "1", "threads", "Images in flight (decode / encode worker threads).", "Runtime"},
    {"dry-run", 0, false, nullptr, "", "Do not reproject; only create the output directory.", "Runtime"},
    {"device", 0, true, "0", "index", "First GPU to use (MI355X addition).", "Runtime"},
    {"gpus", 0, true, "1", "count", "Spread the file list over this many GPUs, contiguous blocks (MI355X addition).", "Runtime"},
    {"streams", 0, true, "0", "count", "Images in flight per GPU: upload, kernel and download of consecutive images overlap (0: one more than the GPU's share of -j, at least 3; MI355X addition).", "Runtime"},
    {"help", 'h', false, nullptr, "", "Show help.", "Runtime"},
};

} // namespace

std::string help_text(const char *argv0) {
  std::string s = "Lens reprojection on MI355X: re-renders images taken through one known lens as seen through\n"
                  "another (rectilinear, equidistant fisheye, equirectangular), with the flag set of\n"
                  "IDLabMedia/image-lens-reproject.\nUsage:\n  ";
  s += argv0;
  s += " [OPTION...]\n";
  const char *group = "";
  for (const Option &o : kOptions) {
    if (std::strcmp(group, o.group)) {
      group = o.group;
      s += std::string("\n ") + group + " options:\n";
    }
    std::string flag = "  ";
    flag += o.short_name ? std::string("-") + o.short_name + ", " : std::string("    ");
    flag += std::string("--") + o.name;
    if (o.takes_value) flag += std::string(" ") + o.arg_help;
    if (flag.size() < 34) flag.resize(34, ' ');
    s += flag + " " + o.help;
    if (o.default_value && *o.default_value) s += std::string(" (default: ") + o.default_value + ")";
    s += "\n";
  }
  return s;
}

namespace {
const Option *find_option(const std::string &name, char short_name) {
  for (const Option &o : kOptions)
    if ((!name.empty() && name == o.name) || (short_name && short_name == o.short_name)) return &o;
  return nullptr;
}

} // namespace

CommandLine parse_command_line(int argc, char **argv) {
  CommandLine a;
  for (const Option &o : kOptions)
    if (o.default_value) a.values[o.name] = o.default_value;
  for (int i = 1; i < argc; ++i) {
    std::string tok = argv[i];
    const Option *o = nullptr;
    std::string inline_value;
    bool has_inline = false;
    if (tok.rfind("--", 0) == 0) {
      std::string name = tok.substr(2);
      const size_t eq = name.find('=');
      if (eq != std::string::npos) {
        inline_value = name.substr(eq + 1);
        name = name.substr(0, eq);
        has_inline = true;
      }
      o = find_option(name, 0);
      if (!o) throw std::invalid_argument("Option '" + name + "' does not exist");
    } else if (tok.size() >= 2 && tok[0] == '-') {
      o = find_option("", tok[1]);
      if (!o) throw std::invalid_argument(std::string("Option '") + tok[1] + "' does not exist");
      if (tok.size() > 2) {
        inline_value = tok.substr(2);
        has_inline = true;
      }
    } else {
      throw std::invalid_argument("Unexpected argument '" + tok + "'");
    }
    a.given.insert(o->name);
    if (o->takes_value) {
      if (has_inline) {
        a.values[o->name] = inline_value;
      } else {
        if (i + 1 >= argc) throw std::invalid_argument(std::string("Option '") + o->name + "' is missing an argument");
        a.values[o->name] = argv[++i];
      }
    } else if (has_inline) {
      throw std::invalid_argument(std::string("Option '") + o->name + "' takes no argument");
    }
  }
  return a;
}


} // namespace lrp_cli
