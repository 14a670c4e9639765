// lrp_cli_options.h — the command line of the `reproject` front end: one table of options (the
// flag set of the reference CLI, README.md:63-157 / src/main.cpp:150-255, plus --device / --gpus)
// that the parser, the defaults and --help are generated from.
#pragma once

#include <map>
#include <set>
#include <string>

namespace lrp_cli {

struct CommandLine {
  std::map<std::string, std::string> values; // defaults, overridden by what was given
  std::set<std::string> given;
  bool has(const std::string &k) const { return given.count(k) != 0; }
  const std::string &operator[](const std::string &k) const { return values.at(k); }
};

// throws std::invalid_argument with a message on a malformed command line
CommandLine parse_command_line(int argc, char **argv);
std::string help_text(const char *argv0);

} // namespace lrp_cli
