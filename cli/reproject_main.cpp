// reproject_main.cpp — command line front end over liblrp_hip.so with the flag surface and
// the per-file worker sequence of the reference CLI (reference src/main.cpp:144-660; the
// flag list is README.md:63-157 there).  Row f2 of SURVEY.md §8f.
//
// Same flags, defaults, validation messages and stdout progress lines; the differences
// are listed in `--help`:
//   * codecs (cli/lrp_image_io.cpp): PNG through libpng, scanline OpenEXR with NO / ZIPS /
//     ZIP compression; JPEG input is not read;
//   * the pixel work runs on the GPU: --device N picks the first GPU, --gpus G spreads the
//     sorted file list over G GPUs in contiguous blocks (no communication: files are
//     independent), -j N worker threads decode / encode and feed the GPUs.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <map>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <fstream>
#include <sstream>

#include "lrp.h"
#include "lrp_config.h"
#include "lrp_image_io.h"
#include "lrp_json.h"

namespace fs = std::filesystem;

namespace {

// ---- command line ---------------------------------------------------------------
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
    {"input-dir", 'i', true, nullptr, "dir", "Directory with the images to reproject (.exr, .png).", "Input/output"},
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
    {"help", 'h', false, nullptr, "", "Show help.", "Runtime"},
};

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
  s += "\nNot built in this version: JPEG input.\n";
  return s;
}

struct Args {
  std::map<std::string, std::string> values;
  std::set<std::string> given;
  int count(const std::string &k) const { return given.count(k) ? 1 : 0; }
  const std::string &operator[](const std::string &k) const { return values.at(k); }
};

const Option *find_option(const std::string &name, char short_name) {
  for (const Option &o : kOptions)
    if ((!name.empty() && name == o.name) || (short_name && short_name == o.short_name)) return &o;
  return nullptr;
}

// throws std::invalid_argument with a message on a malformed command line
Args parse_args(int argc, char **argv) {
  Args a;
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

// ---- lens parsing (reference src/main.cpp:15-95) -----------------------------------
int parse_rectilinear(const std::string &lstr, float res_x, float res_y, lrp_lens &li) {
  const size_t comma = lstr.find(",");
  if (comma == std::string::npos) {
    std::printf("Error: Required format for --rectilinear focal_len,sensor_width\n");
    return 1;
  }
  lrp_lens_rectilinear(&li, (float)std::atof(lstr.substr(0, comma).c_str()), (float)std::atof(lstr.substr(comma + 1).c_str()),
                       res_x, res_y);
  return 0;
}

int parse_equisolid(const std::string &lstr, float res_x, float res_y, lrp_lens &li) {
  const size_t comma1 = lstr.find(",");
  const size_t comma2 = comma1 == std::string::npos ? std::string::npos : lstr.find(",", comma1 + 1);
  if (comma1 == std::string::npos || comma2 == std::string::npos) {
    std::printf("Error: Required format for --equisolid focal_len,sensor_width,fov\n");
    return 1;
  }
  std::memset(&li, 0, sizeof(li));
  li.type = LRP_FISHEYE_EQUISOLID;
  li.u.fisheye_equisolid.focal_length = (float)std::atof(lstr.substr(0, comma1).c_str());
  li.u.fisheye_equisolid.fov = (float)std::atof(lstr.substr(comma2 + 1).c_str());
  li.sensor_width = (float)std::atof(lstr.substr(comma1 + 1, comma2).c_str()); // (sic) src/main.cpp:45
  li.sensor_height = res_y / res_x * li.sensor_width;
  return 0;
}

int parse_equidistant(const std::string &lstr, lrp_lens &li) {
  lrp_lens_equidistant(&li, (float)std::atof(lstr.c_str()));
  return 0;
}

std::vector<std::string> split_commas(const std::string &text) {
  std::vector<std::string> parts;
  size_t begin = 0;
  while (true) {
    const size_t comma = text.find(',', begin);
    parts.push_back(text.substr(begin, comma == std::string::npos ? std::string::npos : comma - begin));
    if (comma == std::string::npos) return parts;
    begin = comma + 1;
  }
}

// "full" or longitude_min,longitude_max,latitude_min,latitude_max in radians (src/main.cpp:58-95)
int parse_equirectangular(const std::string &lstr, lrp_lens &li) {
  if (lstr == "full") {
    lrp_lens_equirectangular_full(&li);
    return 0;
  }
  const std::vector<std::string> parts = split_commas(lstr);
  if (parts.size() != 4) {
    std::printf("Error: expected 4 arguments for equirectangular, got %d.\n", (int)parts.size());
    return 1;
  }
  // parsed as double, narrowed to float on assignment, like the reference
  lrp_lens_equirectangular(&li, (float)std::atof(parts[0].c_str()), (float)std::atof(parts[1].c_str()),
                           (float)std::atof(parts[2].c_str()), (float)std::atof(parts[3].c_str()));
  return 0;
}

// ---- per-run settings -----------------------------------------------------------
struct Job {
  int num_samples = 1, interpolation = LRP_BICUBIC;
  fs::path output_dir;
  double scale = 0.0;
  int ores_x = 0, ores_y = 0;
  lrp_lens input_lens{}, output_lens{};
  float rotation[9];
  bool reproject = true, store_exr = false, store_png = false, skip_if_exists = false;
  double exposure = 1.0, reinhard = 1.0;
};

[[noreturn]] void die_like_reference(const char *msg) {
  std::printf("%s\n", msg);
  std::exit(1);
}

// One file: the worker lambda of reference src/main.cpp:541-620.
void process_file(const Job &job, const fs::path &p, int device, std::atomic_int &done_count, int count) {
  try {
    fs::path output_path_base = job.output_dir / p.filename();
    fs::path output_path_png = output_path_base;
    output_path_png.replace_extension(".png");
    fs::path output_path_exr = output_path_base;
    output_path_exr.replace_extension(".exr");

    bool exists = true;
    if (job.store_png && !fs::exists(output_path_png)) exists = false;
    if (job.store_exr && !fs::exists(output_path_exr)) exists = false;
    if (exists && job.skip_if_exists) {
      std::printf("Skipping '%s'. Already exists.\n", output_path_png.c_str());
      done_count++;
      return;
    }

    lrp_io::Frame input;
    const std::string ext = p.extension().string();
    if (ext == ".exr") {
      input = lrp_io::read_exr(p.string());
    } else if (ext == ".png") {
      input = lrp_io::read_png(p.string());
    } else if (ext == ".jpeg" || ext == ".jpg") { // src/main.cpp:570-571
      input = lrp_io::read_jpeg(p.string());
    } else {
      std::printf("Input format not supported: %s\n", ext.c_str());
      return; // the reference carries on with an uninitialised image here
    }

    lrp_io::Frame output;
    output.width = job.ores_x; // src/main.cpp:581-587: the run's output size, not the file's
    output.height = job.ores_y;
    output.channels = input.channels;
    output.data_layout = input.data_layout;
    if (output.width < 1 || output.height < 1) throw std::runtime_error("empty output image");
    output.data.resize((size_t)output.width * output.height * output.channels);

    const bool post = job.exposure != 1.0 || job.reinhard != 1.0; // :601
    const lrp_post pp{(float)job.exposure, (float)job.reinhard};  // narrowed at the call, :602
    lrp_image in{}, out{};
    in.lens = job.input_lens;
    in.width = input.width;
    in.height = input.height;
    in.channels = input.channels;
    in.data = input.data.data();
    in.data_layout = input.data_layout;
    out.lens = job.output_lens;
    out.width = output.width;
    out.height = output.height;
    out.channels = output.channels;
    out.data = output.data.data();
    out.data_layout = output.data_layout;

    int st = LRP_OK;
    if (!job.reproject && job.scale == 1.0) { // :592-595
      if (input.data.size() < output.data.size()) throw std::runtime_error("input smaller than the configured resolution");
      std::memcpy(output.data.data(), input.data.data(), output.data.size() * sizeof(float));
      if (post) st = lrp_post_process(&out, pp.exposure, pp.reinhard, device);
    } else {
      // reproject() and post_process() in one kernel (same bits as the two calls)
      st = lrp_reproject(&in, &out, job.num_samples, job.interpolation, job.rotation, post ? &pp : nullptr, device);
    }
    if (st == LRP_ERR_OUTPUT_LENS || st == LRP_ERR_INPUT_LENS || st == LRP_ERR_INTERPOLATION)
      die_like_reference(lrp_strerror(st)); // src/reproject.cpp:365-366,396-397,416-417
    if (st != LRP_OK) throw std::runtime_error(std::string(lrp_strerror(st)) + ": " + lrp_last_error());

    if (job.store_png) lrp_io::save_png(output, output_path_png.string());
    if (job.store_exr) lrp_io::save_exr(output, output_path_exr.string());

    const int dc = ++done_count;
    std::printf("%4d / %4d: %s\n", dc, count, p.stem().c_str());
  } catch (const std::exception &e) {
    std::printf("Error: %s\n", e.what());
  }
}

} // namespace

int main(int argc, char **argv) {
  Args result;
  Job job;
  std::string input_single, input_dir;
  int num_threads = 1, device = 0, gpus = 1;
  bool dry_run = false;
  try {
    result = parse_args(argc, argv);
    if (result.count("help")) {
      std::printf("%s\n", help_text(argv[0]).c_str());
      return 0;
    }
    if (result.count("input-dir") && result.count("single")) {
      std::printf("Error: cannot specify both --input-dir and --single.\n");
      std::printf("%s\n", help_text(argv[0]).c_str());
      return 1;
    } else if (result.count("input-dir")) {
      input_dir = result["input-dir"];
    } else if (result.count("single")) {
      input_single = result["single"];
    } else {
      std::printf("Error: No input specified.\n");
      return 1;
    }
    if (!result.count("output-dir")) throw std::invalid_argument("Option 'output-dir' has no value");
    job.output_dir = result["output-dir"];
    job.num_samples = std::atoi(result["samples"].c_str());
    num_threads = std::max(1, std::atoi(result["parallel"].c_str()));
    device = std::atoi(result["device"].c_str());
    gpus = std::max(1, std::atoi(result["gpus"].c_str()));
    if (result.count("output-resolution")) {
      const std::string arg = result["output-resolution"];
      const size_t comma = arg.find(",");
      if (comma == std::string::npos || comma == arg.length() - 1 || comma == 0) {
        std::printf("Error: Specify both width and height, separated by a comma in output-resolution.\n");
        return 1;
      }
      job.ores_x = std::atoi(arg.substr(0, comma).c_str());
      job.ores_y = std::atoi(arg.substr(comma + 1).c_str());
    } else {
      job.scale = std::atof(result["scale"].c_str());
    }
    {
      // src/main.cpp:312-325: degrees -> radians in double, narrowed to float
      const std::string euler = result["rotation"];
      const size_t comma0 = euler.find(',');
      const size_t comma1 = euler.find(',', comma0 + 1);
      const float pan = (float)(std::atof(euler.substr(0, comma0).c_str()) / 180.0 * M_PI);
      const float pitch = (float)(std::atof(euler.substr(comma0 + 1, comma1).c_str()) / 180.0 * M_PI);
      const float roll = (float)(std::atof(euler.substr(comma1 + 1).c_str()) / 180.0 * M_PI);
      lrp_rotation_matrix(pan, pitch, roll, job.rotation);
    }
    job.exposure = std::pow(2.0, std::atof(result["exposure"].c_str()));
    job.reinhard = std::atof(result["reinhard"].c_str());
    if (result.count("no-reproject")) job.reproject = false;
  } catch (const std::invalid_argument &e) {
    std::printf("%s\n\n%s\n", e.what(), help_text(argv[0]).c_str());
    return 1;
  }
  if (result.count("dry-run")) dry_run = true;
  if (result.count("skip-if-exists")) job.skip_if_exists = true;
  job.store_exr = result.count("exr");
  job.store_png = result.count("png");
  if (!job.store_exr && !job.store_png) {
    std::printf("Error: Did not specify any output format.\nChoose --png or --exr. (both are possible).\n");
    return 1;
  }

  int found_interpolation_flag = 0;
  if (result.count("nn")) {
    found_interpolation_flag++;
    job.interpolation = LRP_NEAREST;
  }
  if (result.count("bl")) {
    found_interpolation_flag++;
    job.interpolation = LRP_BILINEAR;
  }
  if (result.count("bc")) {
    found_interpolation_flag++;
    job.interpolation = LRP_BICUBIC;
  }
  if (found_interpolation_flag > 1) { // the reference warns and carries on with the last one (src/main.cpp:373-376)
    std::printf("Cannot specify more than one interpolation method.\n\n");
    std::printf("%s", help_text(argv[0]).c_str());
  }
  const std::string filter_prefix = result["filter-prefix"], filter_suffix = result["filter-suffix"];

  int ires_x = 0, ires_y = 0;
  lrp_json::Value out_cfg;
  std::string output_cfg_file;
  if (result.count("no-configs")) {
    const std::string lstr = result["no-configs"];
    const size_t comma = lstr.find(",");
    ires_x = std::atoi(lstr.substr(0, comma).c_str());
    ires_y = comma == std::string::npos ? 0 : std::atoi(lstr.substr(comma + 1).c_str());
    int input_lens_types_found = 0;
    if (result.count("i-rectilinear")) {
      if (parse_rectilinear(result["i-rectilinear"], (float)ires_x, (float)ires_y, job.input_lens)) return 1;
      input_lens_types_found++;
    }
    if (result.count("i-equisolid")) {
      if (parse_equisolid(result["i-equisolid"], (float)ires_x, (float)ires_y, job.input_lens)) return 1;
      input_lens_types_found++;
    }
    if (result.count("i-equidistant")) {
      if (parse_equidistant(result["i-equidistant"], job.input_lens)) return 1;
      input_lens_types_found++;
    }
    if (result.count("i-equirectangular")) {
      if (parse_equirectangular(result["i-equirectangular"], job.input_lens)) return 1;
      input_lens_types_found++;
    }
    if (input_lens_types_found > 1) {
      std::printf("Error: only specify one input lens type: [--i-rectilinear, --i-equisolid, --i-equidistant, "
                  "--i-equirectangular].\n");
      return 1;
    }
  } else {
    // config-file mode, src/main.cpp:429-443
    if (!result.count("input-cfg") || !result.count("output-cfg")) {
      std::printf("Option 'input-cfg' / 'output-cfg' has no value (or use --no-configs width,height)\n\n%s\n",
                  help_text(argv[0]).c_str());
      return 1;
    }
    output_cfg_file = result["output-cfg"];
    try {
      std::ifstream in(result["input-cfg"]);
      if (!in) throw std::invalid_argument("cannot open " + result["input-cfg"]);
      std::stringstream buf;
      buf << in.rdbuf();
      out_cfg = lrp_json::parse(buf.str());
      std::printf("Found camera config: %s\n", out_cfg.at("camera").dump(1).c_str());
      ires_x = out_cfg.at("resolution").at(0).as_int();
      ires_y = out_cfg.at("resolution").at(1).as_int();
      job.input_lens = lrp_cfg::extract_lens_info_from_config(out_cfg);
    } catch (const std::exception &e) { // the reference lets the exception end the program
      std::printf("Error: %s\n", e.what());
      return 1;
    }
  }

  int output_lens_types_found = 0;
  if (job.ores_x == 0 && job.ores_y == 0) { // src/main.cpp:448-451
    job.ores_x = int(ires_x * job.scale);
    job.ores_y = int(ires_y * job.scale);
  }
  if (result.count("rectilinear")) {
    if (parse_rectilinear(result["rectilinear"], (float)job.ores_x, (float)job.ores_y, job.output_lens)) return 1;
    output_lens_types_found++;
  }
  if (result.count("equisolid")) {
    if (parse_equisolid(result["equisolid"], (float)job.ores_x, (float)job.ores_y, job.output_lens)) return 1;
    output_lens_types_found++;
  }
  if (result.count("equidistant")) {
    if (parse_equidistant(result["equidistant"], job.output_lens)) return 1;
    output_lens_types_found++;
  }
  if (result.count("equirectangular")) {
    if (parse_equirectangular(result["equirectangular"], job.output_lens)) return 1;
    output_lens_types_found++;
  }
  if (!job.reproject) {
    job.output_lens = job.input_lens;
    output_lens_types_found++;
  }
  if (output_lens_types_found > 1) {
    std::printf("Error: only specify one output lens type: [--rectilinear, --equisolid, --equidistant, "
                "--equirectangular, --no-reproject].\n");
    return 1;
  }

  std::printf("Creating directory: %s\n", job.output_dir.c_str());
  std::error_code ec;
  fs::create_directory(job.output_dir, ec);

  if (!result.count("no-configs")) {
    // src/main.cpp:497-529: the input config with the output lens, the output resolution and
    // the frame list filtered like the files
    try {
      lrp_cfg::store_lens_info_in_config(job.output_lens, out_cfg);
      out_cfg["resolution"][0] = lrp_json::Value::integer(job.ores_x);
      out_cfg["resolution"][1] = lrp_json::Value::integer(job.ores_y);
      if (out_cfg.contains("frames")) {
        auto &frames = out_cfg["frames"].arr;
        for (size_t i = 0; i < frames.size();) {
          const std::string name = frames[i].at("name").str();
          bool remove = false;
          if (name.size() < filter_prefix.size() || name.size() < filter_suffix.size())
            remove = true;
          else if (name.substr(0, filter_prefix.size()) != filter_prefix)
            remove = true;
          else if (name.substr(name.size() - filter_suffix.size()) != filter_suffix)
            remove = true;
          if (remove)
            frames.erase(frames.begin() + (long)i);
          else
            ++i;
        }
      }
      std::printf("Saving output config: %s\n", output_cfg_file.c_str());
      std::ofstream out(output_cfg_file);
      out << out_cfg.dump(2);
    } catch (const std::exception &e) {
      std::printf("Error: %s\n", e.what());
      return 1;
    }
  }

  if (dry_run) {
    std::printf("Dry-run. Exiting.\n");
    return 0;
  }

  // ---- the file list (src/main.cpp:624-655) ------------------------------------
  std::vector<fs::path> files;
  if (!input_dir.empty()) {
    std::vector<fs::path> paths;
    for (const auto &entry : fs::directory_iterator(fs::path(input_dir)))
      if (entry.is_regular_file()) paths.push_back(entry.path());
    std::sort(paths.begin(), paths.end());
    for (const fs::path &p : paths) {
      const std::string fn = p.filename().string();
      if (fn.size() < filter_prefix.size() || fn.size() < filter_suffix.size()) continue;
      if (fn.substr(0, filter_prefix.size()) != filter_prefix) continue;
      if (fn.substr(fn.size() - filter_suffix.size()) != filter_suffix) continue;
      if (p.extension() == ".exr" || p.extension() == ".png") files.push_back(p);
    }
  } else {
    files.push_back(fs::path(input_single));
  }

  // ---- workers: contiguous block of the sorted list per GPU, -j threads in total ------
  // A pure copy run (--no-reproject, scale 1, no colour processing: src/main.cpp:592-595)
  // touches no pixel arithmetic and needs no GPU; everything else does.
  const bool needs_gpu = job.reproject || job.scale != 1.0 || job.exposure != 1.0 || job.reinhard != 1.0;
  const int n_dev = lrp_device_count();
  if (needs_gpu && n_dev < 1) {
    std::printf("Error: no usable HIP device (this build has no CPU path).\n");
    return 1;
  }
  gpus = std::min(gpus, std::max(1, n_dev - device));
  const int count = (int)files.size();
  std::atomic_int done_count{0};
  std::atomic_size_t next_in_block[64];
  const size_t per = gpus > 0 ? (files.size() + (size_t)gpus - 1) / (size_t)gpus : files.size();
  for (int g = 0; g < 64; ++g) next_in_block[g] = 0;
  const int workers = std::max(num_threads, gpus);
  std::vector<std::thread> pool;
  for (int t = 0; t < workers; ++t) {
    pool.emplace_back([&, t] {
      const int g = t % gpus; // this worker's GPU and file block
      const size_t begin = std::min(files.size(), (size_t)g * per), end = std::min(files.size(), begin + per);
      for (;;) {
        const size_t i = begin + next_in_block[g].fetch_add(1);
        if (i >= end) break;
        process_file(job, files[i], device + g, done_count, count);
      }
    });
  }
  for (auto &th : pool) th.join();
  return 0;
}
