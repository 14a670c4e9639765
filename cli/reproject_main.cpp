// reproject_main.cpp — the `reproject` command: the reference CLI's flags, defaults, validation
// messages and progress lines (reference src/main.cpp:144-660, README.md:63-157 there; row f2 of
// SURVEY.md section 8f) in front of liblrp_hip.so.
//
//   lrp_cli_options   the option table: parser, defaults and --help come from it
//   lrp_run_plan      command line -> RunPlan (lenses, sizes, rotation, colour, outputs), the output
//                     config document, the file list
//   lrp_engine        RunPlan x files -> images: host threads decode / encode, the GPU converts pixel
//                     formats, reprojects and tonemaps; contiguous blocks of the file list per GPU
//
// Differences from the reference, also listed by --help: the pixel work runs on the GPU (--device N,
// --gpus G, -j N worker threads); per-file errors make the exit status 1 at the end of the run.
#include <cstdio>
#include <filesystem>
#include <stdexcept>

#include "lrp_cli_options.h"
#include "lrp_engine.h"
#include "lrp_run_plan.h"

int main(int argc, char **argv) {
  using namespace lrp_cli;
  CommandLine cl;
  try {
    cl = parse_command_line(argc, argv);
  } catch (const std::invalid_argument &e) {
    std::printf("%s\n\n%s\n", e.what(), help_text(argv[0]).c_str());
    return 1;
  }
  if (cl.has("help")) {
    std::printf("%s\n", help_text(argv[0]).c_str());
    return 0;
  }
  RunPlan plan;
  if (int rc = resolve_run_plan(cl, argv[0], plan)) return rc;

  std::printf("Creating directory: %s\n", plan.output_dir.c_str());
  std::error_code ec;
  std::filesystem::create_directory(plan.output_dir, ec);
  if (plan.has_config)
    if (int rc = write_output_config(plan)) return rc;
  if (plan.dry_run) {
    std::printf("Dry-run. Exiting.\n");
    return 0;
  }
  const RunResult r = run_files(plan, enumerate_inputs(plan));
  return (r.aborted || r.failed > 0) ? 1 : 0;
}
