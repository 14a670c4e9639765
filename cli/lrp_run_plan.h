// lrp_run_plan.h — everything a run is made of, resolved from the command line before any pixel
// is touched: what to read, what lens the inputs have, what to render (lens, size, rotation,
// sampling, colour processing), what to write.  The values and their quirks are the reference
// CLI's (src/main.cpp:256-535); how they are derived is a sequence of small resolvers over the
// option table (lrp_run_plan.cpp).
#pragma once

#include <filesystem>
#include <string>
#include <vector>

#include "lrp.h"
#include "lrp_cli_options.h"
#include "lrp_json.h"

namespace lrp_cli {

struct RunPlan {
  // inputs
  std::string input_dir, input_single; // exactly one is set
  std::string filter_prefix, filter_suffix;
  int in_width = 0, in_height = 0; // of the run (--no-configs or the input config), not of a file
  lrp_lens input_lens{};
  // rendering
  lrp_lens output_lens{};
  int out_width = 0, out_height = 0;
  double scale = 0.0;
  bool reproject = true; // false: --no-reproject (the output lens is the input lens)
  int num_samples = 1, interpolation = LRP_BICUBIC;
  float rotation[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double exposure = 1.0, reinhard = 1.0;
  // outputs
  std::filesystem::path output_dir;
  bool write_exr = false, write_png = false, skip_if_exists = false, dry_run = false;
  // config-file mode: the document to write next to the images, and where
  bool has_config = false;
  lrp_json::Value config;
  std::string output_config_path;
  // execution
  int workers = 1, first_device = 0, gpus = 1;
  int streams = 0; // image slots of a GPU's pipeline (--streams; 0: automatic)

  bool post_process() const { return exposure != 1.0 || reinhard != 1.0; } // src/main.cpp:601
  bool copies_pixels() const { return !reproject && scale == 1.0; }        // :592-595
};

// Fills `plan`.  Returns 0, or the process exit code after having printed the reference's message.
int resolve_run_plan(const CommandLine &cl, const char *argv0, RunPlan &plan);
// Config-file mode: the input config with the output lens, the output resolution and the frame
// list filtered like the files (src/main.cpp:497-529).  Returns 0 or an exit code.
int write_output_config(RunPlan &plan);
// The sorted, filtered file list (src/main.cpp:624-655).
std::vector<std::filesystem::path> enumerate_inputs(const RunPlan &plan);
bool name_passes_filters(const RunPlan &plan, const std::string &name);

} // namespace lrp_cli
