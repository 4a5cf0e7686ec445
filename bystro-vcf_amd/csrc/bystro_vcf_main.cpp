// bystro-vcf — stdin VCF -> stdout TSV, the reference's process surface (main.go:82-217) over libbvcf.
//
// Flag names, defaults and the order of optional output columns are the reference's
// (setup(), main.go:84-99).  Go's `flag` accepts -x and --x, "--x=v" and "--x v"; bools take
// presence or =true/false; parsing stops at the first non-flag argument.
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "../../include/bvcf.h"

namespace {

struct Cli {
  std::string in, out, err, dosage, sample, fam, empty = "!", delim = ";", cpu_profile;
  std::string allow = "PASS,.", exclude;
  bool no_out = false, keep_id = false, keep_qual = false, keep_pos = false, keep_info = false;
  // extension: a comma-separated list of HIP device ordinals, or "all" (every visible device).  The default is ONE
  // device: a tool that is dropped into a pipeline on a shared node must not take GPUs it was not given.
  std::string devices = "0";
  unsigned long long batch_mb = 0;
};

bool parse_bool(const char *v, bool *out) {
  // strconv.ParseBool
  static const char *t[] = {"1", "t", "T", "TRUE", "true", "True"};
  static const char *f[] = {"0", "f", "F", "FALSE", "false", "False"};
  for (auto s : t)
    if (!strcmp(v, s)) return *out = true, true;
  for (auto s : f)
    if (!strcmp(v, s)) return *out = false, true;
  return false;
}

int parse(int argc, char **argv, Cli &c) {
  struct B {
    const char *name;
    bool *v;
  } bools[] = {{"noOut", &c.no_out}, {"keepId", &c.keep_id}, {"keepQual", &c.keep_qual},
               {"keepPos", &c.keep_pos}, {"keepInfo", &c.keep_info}};
  struct S {
    const char *name;
    std::string *v;
  } strs[] = {{"in", &c.in}, {"fam", &c.fam}, {"err", &c.err}, {"out", &c.out}, {"dosageOutput", &c.dosage},
              {"sample", &c.sample}, {"emptyField", &c.empty}, {"fieldDelimiter", &c.delim},
              {"cpuProfile", &c.cpu_profile}, {"allowFilter", &c.allow}, {"excludeFilter", &c.exclude}};
  for (int i = 1; i < argc; i++) {
    const char *a = argv[i];
    if (a[0] != '-' || !a[1]) break;  // first non-flag ends parsing
    a++;
    if (*a == '-') a++;
    if (!*a) break;  // "--" terminator
    std::string name(a);
    std::string val;
    bool has_val = false;
    size_t eq = name.find('=');
    if (eq != std::string::npos) {
      val = name.substr(eq + 1);
      name = name.substr(0, eq);
      has_val = true;
    }
    bool done = false;
    for (auto &b : bools)
      if (name == b.name) {
        if (!has_val)
          *b.v = true;
        else if (!parse_bool(val.c_str(), b.v)) {
          fprintf(stderr, "invalid boolean value \"%s\" for -%s: parse error\n", val.c_str(), name.c_str());
          return 2;
        }
        done = true;
      }
    if (done) continue;
    for (auto &s : strs)
      if (name == s.name) {
        if (!has_val) {
          if (i + 1 >= argc) {
            fprintf(stderr, "flag needs an argument: -%s\n", name.c_str());
            return 2;
          }
          val = argv[++i];
        }
        *s.v = val;
        done = true;
      }
    if (done) continue;
    // extensions of this build (not in the reference): the devices the blocks are dealt to (the
    // counterpart of the reference's NumCPU workers), the block size
    if (name == "devices" || name == "device" || name == "batchMB") {
      if (!has_val) {
        if (i + 1 >= argc) {
          fprintf(stderr, "flag needs an argument: -%s\n", name.c_str());
          return 2;
        }
        val = argv[++i];
      }
      if (name == "batchMB")
        c.batch_mb = strtoull(val.c_str(), nullptr, 10);
      else
        c.devices = val;
      continue;
    }
    fprintf(stderr, "flag provided but not defined: -%s\n", name.c_str());
    return 2;
  }
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  Cli c;
  int rc = parse(argc, argv, c);
  if (rc) return rc;

  int fd_in = 0, fd_out = 1, fd_err = 2;
  if (!c.in.empty()) {  // main.go:138-147
    fd_in = open(c.in.c_str(), O_RDONLY);
    if (fd_in < 0) {
      fprintf(stderr, "open %s: %s\n", c.in.c_str(), strerror(errno));
      return 1;
    }
  }
  if (!c.err.empty()) {  // main.go:150-156 (the reference opens this read-only; we open it for append)
    fd_err = open(c.err.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd_err < 0) {
      fprintf(stderr, "open %s: %s\n", c.err.c_str(), strerror(errno));
      return 1;
    }
  }
  if (c.no_out && !c.out.empty()) {  // main.go:160-162
    dprintf(fd_err, "Cannot specify --noOut and --out\n");
    return 1;
  }
  if (c.no_out && c.dosage.empty()) {  // main.go:164-166
    dprintf(fd_err, "When specifying --noOut, must specify --dosageOutput\n");
    return 1;
  }
  if (!c.out.empty()) {  // main.go:172
    fd_out = open(c.out.c_str(), O_WRONLY | O_CREAT, 0644);
    if (fd_out < 0) {
      dprintf(fd_err, "open %s: %s\n", c.out.c_str(), strerror(errno));
      return 1;
    }
  }

  bvcf_config cfg;
  bvcf_config_defaults(&cfg);
  cfg.empty_field = c.empty.c_str();
  cfg.field_delimiter = c.delim.c_str();
  cfg.allow_filter = c.allow.c_str();
  cfg.exclude_filter = c.exclude.c_str();
  cfg.keep_id = c.keep_id;
  cfg.keep_info = c.keep_info;
  cfg.keep_pos = c.keep_pos;
  cfg.keep_qual = c.keep_qual;
  // --devices all: every visible HIP device; a device only gets a ctx once a block reaches it
  std::vector<int32_t> devs;
  if (c.devices == "all") {
    const int n = bvcf_device_count();
    for (int d = 0; d < n; d++) devs.push_back(d);
  } else {
    const char *p = c.devices.c_str();
    while (*p) {
      char *e = nullptr;
      const long d = strtol(p, &e, 10);
      if (e == p || d < 0) {
        dprintf(fd_err, "invalid value \"%s\" for flag -devices\n", c.devices.c_str());
        return 2;
      }
      devs.push_back((int32_t)d);
      p = *e == ',' ? e + 1 : e;
      if (*e && *e != ',') {
        dprintf(fd_err, "invalid value \"%s\" for flag -devices\n", c.devices.c_str());
        return 2;
      }
    }
  }
  if (devs.empty()) {
    // no device: bvcf_run_fd reports it (there is no CPU path)
    devs.push_back(0);
  }
  cfg.device = devs[0];
  cfg.devices = devs.data();
  cfg.n_devices = (uint32_t)devs.size();
  if (c.batch_mb) cfg.max_batch_bytes = c.batch_mb << 20;
  cfg.sample_list_path = c.sample.c_str();
  cfg.dosage_path = c.dosage.c_str();  // main.go:89
  cfg.no_out = c.no_out;               // main.go:88
  const char *raw = getenv("BVCF_RAW_SAMPLE_NAMES");
  if (raw && *raw == '1') cfg.normalize_header = 0;

  // this process ends with the run: unpinning 0.5 GB of buffers and tearing down the HIP runtime only delays the exit
  cfg.leave_teardown_to_exit = 1;
  uint64_t n_lines = 0;
  rc = bvcf_run_fd(&cfg, fd_in, fd_out, fd_err, &n_lines);
  // (a write error the file system reports late -- quota, a network file system -- shows up at close)
  if (fd_out != 1 && close(fd_out) != 0 && rc == BVCF_OK) {
    dprintf(fd_err, "close %s: %s\n", c.out.c_str(), strerror(errno));
    rc = BVCF_E_IO;
  }
  fflush(nullptr);
  // (a profiler writes its results from exit handlers: leave normally under rocprofv3)
  const char *pre = getenv("LD_PRELOAD");
  if (getenv("ROCP_TOOL_LIBRARIES") || (pre && strstr(pre, "rocprof"))) return rc == BVCF_OK ? 0 : 1;
  _exit(rc == BVCF_OK ? 0 : 1);  // log.Fatal exits 1
}
