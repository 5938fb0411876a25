// config_probe.cpp -- CPU test driver (built with -fsanitize=address,undefined) for the
// JSON reader and the VPC configuration model: parse, validate, plan; never touches HIP.
//   config_probe FILE...   -> one line per file: rc, path, message
#include <cstdio>
#include <string>

#include "mpc_config.h"

int main(int argc, char **argv)
{
  for (int i = 1; i < argc; i++) {
    std::string text, err;
    if (!mpc::read_file(argv[i], text)) { printf("%s: unreadable\n", argv[i]); continue; }
    mpc::VpcConfig cfg;
    int rc = mpc::parse_vpc_config(text, cfg, err);
    if (rc == 0) {
      mpc::VpcPlan plan;
      mpc::build_vpc_plan(cfg, plan);
      printf("%s: rc 0 L %d M %d path %s tab %zu gtab %zu\n", argv[i], cfg.L, cfg.M, plan.fast ? "fast" : "generic",
             plan.tab.size(), plan.gtab.size());
    } else {
      printf("%s: rc %d %s\n", argv[i], rc, err.c_str());
    }
  }
  return 0;
}
