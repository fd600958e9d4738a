// Command-line options in the style of the reference's ParseOptions
// (/root/reference/src/util/parse-options.h:36-128): --name=value flags registered by
// pointer, --config=FILE, --help, --print-args, --verbose, then positional arguments.
#ifndef KALDI_AMD_CSRC_HOST_PARSE_OPTIONS_H_
#define KALDI_AMD_CSRC_HOST_PARSE_OPTIONS_H_

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "kaldi_io.h"

namespace kaldi_amd {

class ParseOptions {
 public:
  explicit ParseOptions(const char *usage) : usage_(usage) {
    Register("verbose", &verbose_, "Verbose level (higher->more logging)");
    Register("print-args", &print_args_, "Print the command line arguments (to stderr)");
  }
  void Register(const std::string &name, bool *v, const std::string &doc) { Add(name, doc, Opt{kBool, v}); }
  void Register(const std::string &name, int32 *v, const std::string &doc) { Add(name, doc, Opt{kInt, v}); }
  void Register(const std::string &name, float *v, const std::string &doc) { Add(name, doc, Opt{kFloat, v}); }
  void Register(const std::string &name, std::string *v, const std::string &doc) { Add(name, doc, Opt{kString, v}); }

  int Read(int argc, const char *const *argv) {
    std::string line;
    for (int i = 0; i < argc; ++i) line += std::string(argv[i]) + " ";
    int i = 1;
    for (; i < argc; ++i) {
      std::string a = argv[i];
      if (a == "--") { ++i; break; }
      if (a.compare(0, 2, "--") != 0) break;
      if (a == "--help") { PrintUsage(); std::exit(0); }
      std::string key = a.substr(2), val;
      const size_t eq = key.find('=');
      bool has_val = eq != std::string::npos;
      if (has_val) { val = key.substr(eq + 1); key = key.substr(0, eq); }
      if (key == "config") { ReadConfigFile(val); continue; }
      Set(key, val, has_val);
    }
    for (; i < argc; ++i) positional_.push_back(argv[i]);
    if (print_args_) std::cerr << line << "\n";
    return i;
  }
  int NumArgs() const { return static_cast<int>(positional_.size()); }
  std::string GetArg(int i) const {  // 1-based, as in the reference
    if (i < 1 || i > NumArgs()) KALDI_AMD_ERR("ParseOptions::GetArg, invalid index " << i);
    return positional_[i - 1];
  }
  void PrintUsage() const {
    std::cerr << "\n" << usage_ << "\nOptions:\n";
    for (const auto &kv : docs_) std::cerr << "  --" << kv.first << " : " << kv.second << "\n";
    std::cerr << "\n";
  }
  int32 Verbose() const { return verbose_; }

 private:
  enum Kind { kBool, kInt, kFloat, kString };
  struct Opt { Kind kind; void *ptr; };
  void Add(const std::string &name, const std::string &doc, Opt o) { opts_[name] = o; docs_[name] = doc; }
  void Set(const std::string &key, const std::string &val, bool has_val) {
    auto it = opts_.find(key);
    if (it == opts_.end()) KALDI_AMD_ERR("Invalid option --" << key);
    switch (it->second.kind) {
      case kBool:
        if (!has_val || val == "true" || val == "t" || val == "1" || val.empty()) *static_cast<bool *>(it->second.ptr) = true;
        else if (val == "false" || val == "f" || val == "0") *static_cast<bool *>(it->second.ptr) = false;
        else KALDI_AMD_ERR("Invalid format for boolean argument [expected true or false]: " << val);
        break;
      case kInt: {
        char *end = nullptr;
        const long v = strtol(val.c_str(), &end, 10);
        if (val.empty() || *end) KALDI_AMD_ERR("Invalid integer option \"" << val << "\"");
        *static_cast<int32 *>(it->second.ptr) = static_cast<int32>(v);
        break;
      }
      case kFloat: {
        char *end = nullptr;
        const float v = strtof(val.c_str(), &end);
        if (val.empty() || *end) KALDI_AMD_ERR("Invalid floating-point option \"" << val << "\"");
        *static_cast<float *>(it->second.ptr) = v;
        break;
      }
      case kString:
        *static_cast<std::string *>(it->second.ptr) = val;
        break;
    }
  }
  void ReadConfigFile(const std::string &path) {
    std::ifstream f(path);
    if (!f.is_open()) KALDI_AMD_ERR("Cannot open config file: " << path);
    std::string line;
    while (std::getline(f, line)) {
      const size_t hash = line.find('#');
      if (hash != std::string::npos) line = line.substr(0, hash);
      const size_t b = line.find_first_not_of(" \t\r"), e = line.find_last_not_of(" \t\r");
      if (b == std::string::npos) continue;
      line = line.substr(b, e - b + 1);
      if (line.compare(0, 2, "--") != 0) KALDI_AMD_ERR("Reading config file " << path << ": line has no --: " << line);
      std::string key = line.substr(2), val;
      const size_t eq = key.find('=');
      const bool has_val = eq != std::string::npos;
      if (has_val) { val = key.substr(eq + 1); key = key.substr(0, eq); }
      Set(key, val, has_val);
    }
  }
  std::string usage_;
  std::map<std::string, Opt> opts_;
  std::map<std::string, std::string> docs_;
  std::vector<std::string> positional_;
  int32 verbose_ = 0;
  bool print_args_ = true;
};

}  // namespace kaldi_amd

#endif  // KALDI_AMD_CSRC_HOST_PARSE_OPTIONS_H_
