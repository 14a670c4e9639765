// lrp_engine.h — runs a plan over a list of files: -j worker threads decode and encode on the
// host; the pixels cross PCIe in their FILE format (RGBA8 / RGB8 / binary16, page-locked buffers)
// and are converted, reprojected and converted back on the GPU through one three-stage
// lrp_context per device (upload | kernels | download overlap across images).  The sorted file
// list is cut into contiguous blocks, block g -> GPU first_device + g: no
// communication, the reference's one-file-per-pool-thread split (src/main.cpp:538-544).
#pragma once

#include <filesystem>
#include <vector>

#include "lrp_run_plan.h"

namespace lrp_cli {

struct RunResult {
  int failed = 0;        // files that ended in "Error: ..."
  bool aborted = false;  // a dispatch error of reproject() (unsupported lens / interpolation): exit code 1
};

RunResult run_files(const RunPlan &plan, const std::vector<std::filesystem::path> &files);

} // namespace lrp_cli
