// main.cpp -- the `compressor` command line of the reference (src/main.cpp), kept
// flag-for-flag:  compressor -a ALGO -i TRACE [-c CONFIG.json] [-o OUTDIR] [-h]
// stdout "comp.ratio: <double>", CSV rows appended to OUTDIR/<stem>_results.csv and
// OUTDIR/<stem>_results_detail.csv.  This build evaluates VPC, BDI, FPC and BPC (the hot path
// of SURVEY.md section 8) on the MI355X; traces are .npy files, GPGPU-Sim .log files
// (GLOBAL_ACC_R / GLOBAL_ACC_W requests, reference main.cpp:222-224) or APSim .txt files
// (32-byte data beats of handshaking channels).  The other algorithm names are recognised
// and reported as not part of this build.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "BDI.h"
#include "BPC.h"
#include "FPC.h"
#include "LoaderAPSim.h"
#include "LoaderGPGPU.h"
#include "LoaderNPY.h"
#include "VPC.h"
#include "utils.h"

#define REQ_SIZE 32   // line size asked of the APSim loader (reference main.cpp:23)

static comp::CompResult *compressLines(comp::Compressor *compressor, trace::Loader *loader, bool perLine);

static const char *kHelp =
    "Usage:\n"
    "  Compressor [OPTION...]\n"
    "\n"
    "  -a, --algorithm arg  Compression algorithm\n"
    "                       [VPC/FPC/BDI/BPC/CPACK/SC2/PATTERN/VIEWER]. Default=VPC\n"
    "  -i, --input arg      Input GPGPU-Sim trace file path. Supported extensions:\n"
    "                       .log, .npy\n"
    "  -c, --config arg     Config file path (.json).\n"
    "  -o, --output arg     Output directory path\n"
    "  -h, --help           Print usage\n";

struct Args {
  std::string algorithm, input, config, output;
  bool has_algorithm = false, has_input = false, has_config = false, has_output = false, help = false;
  bool per_line = false;   // ADDITIVE --per-line: the reference's loop (GetCacheline -> CompressLine per line) instead of batches
  unsigned long long line_buffer = 0;   // ADDITIVE --line-buffer N: that loop with Compressor::SetLineBuffering(N)
};

static bool take_value(int argc, char **argv, int &i, const std::string &arg, const char *shortf, const char *longf,
                       std::string &out, bool &seen)
{
  const std::string lf = std::string("--") + longf;
  if (arg == shortf || arg == lf) {
    if (i + 1 >= argc) {
      std::cout << "Option '" << longf << "' is missing an argument" << std::endl;
      exit(1);
    }
    out = argv[++i];
    seen = true;
    return true;
  }
  if (arg.compare(0, lf.size() + 1, lf + "=") == 0) {
    out = arg.substr(lf.size() + 1);
    seen = true;
    return true;
  }
  if (arg.size() > 2 && arg.compare(0, 2, shortf) == 0) {   // -aVPC
    out = arg.substr(2);
    seen = true;
    return true;
  }
  return false;
}

int main(int argc, char **argv)
{
  Args a;
  for (int i = 1; i < argc; i++) {
    const std::string arg = argv[i];
    if (arg == "-h" || arg == "--help") { a.help = true; continue; }
    if (arg == "--per-line") { a.per_line = true; continue; }
    if (arg == "--line-buffer") {
      if (i + 1 >= argc) {
        std::cout << "Option 'line-buffer' is missing an argument" << std::endl;
        exit(1);
      }
      a.per_line = true;
      a.line_buffer = strtoull(argv[++i], nullptr, 10);
      continue;
    }
    if (take_value(argc, argv, i, arg, "-a", "algorithm", a.algorithm, a.has_algorithm)) continue;
    if (take_value(argc, argv, i, arg, "-i", "input", a.input, a.has_input)) continue;
    if (take_value(argc, argv, i, arg, "-c", "config", a.config, a.has_config)) continue;
    if (take_value(argc, argv, i, arg, "-o", "output", a.output, a.has_output)) continue;
    std::cout << "Option '" << arg << "' does not exist" << std::endl;
    return 1;
  }
  std::string algorithm = a.has_algorithm ? a.algorithm : "VPC";
  bool help = a.help;
  if (!a.has_input) help = true;
  if (algorithm == "VPC" && !a.has_config) help = true;
  if (help) {
    std::cout << kHelp << std::endl;
    return 0;
  }
  const std::string tracePath = a.input, configPath = a.config, outputDirPath = a.has_output ? a.output : "";

  // loader by extension (reference main.cpp:74-83)
  trace::Loader *loader = nullptr;
  if (mpctext::ends_with(tracePath, ".npy")) {
    loader = new trace::LoaderNPY(tracePath);
  } else if (mpctext::ends_with(tracePath, ".log")) {
    loader = new trace::gpgpusim::LoaderGPGPU(tracePath);
  } else if (mpctext::ends_with(tracePath, ".txt")) {
    loader = new trace::apsim::LoaderGPGPU(tracePath, REQ_SIZE);
  } else {
    std::cerr << "Unsupported extension." << std::endl;
    abort();
  }

  // compressor by name (reference main.cpp:85-127)
  const unsigned lineSize = loader->GetCachelineSize();
  comp::Compressor *compressor = nullptr;
  if (algorithm == "VPC") {
    compressor = new comp::VPC(configPath);
  } else if (algorithm == "BDI") {
    compressor = new comp::BDI(lineSize);
  } else if (algorithm == "FPC") {
    compressor = new comp::FPC(lineSize);
  } else if (algorithm == "BPC") {
    compressor = new comp::BPC(lineSize);
  } else if (algorithm == "CPACK" || algorithm == "SC2" || algorithm == "PATTERN" || algorithm == "VIEWER") {
    std::cout << "Algorithm " << algorithm << " is not part of this build: VPC, BDI, FPC and BPC are (see DESIGN.md, \"Out of scope\")."
              << std::endl;
    return 1;
  } else {
    std::cerr << "Invalid name of algorithm." << std::endl;
    abort();
  }

  // result files (reference main.cpp:129-136)
  const std::string saveFileName = (algorithm == "VPC") ? parseConfig(configPath) : algorithm;
  const std::string compOutputSavePath = outputDirPath + "/" + saveFileName + "_results.csv";
  const std::string compDetailedOutputSavePath = outputDirPath + "/" + saveFileName + "_results_detail.csv";

  if (a.line_buffer) compressor->SetLineBuffering(a.line_buffer);
  comp::CompResult *compStat = compressLines(compressor, loader, a.per_line);

  // workload name = <parent directory>_<file stem> (reference main.cpp:141-157)
  std::string workloadName;
  {
    std::vector<std::string> parts = mpctext::split(tracePath, "/");
    if (parts.size() < 2) {
      std::cout << "The trace path needs at least one '/' (workload name = <directory>_<file>)." << std::endl;
      return 1;
    }
    std::string benchmarkName = parts[parts.size() - 2];
    std::string appName = parts[parts.size() - 1];
    mpctext::replace_all(appName, ".log", "");
    mpctext::replace_all(appName, ".npy", "");
    mpctext::replace_all(appName, ".txt", "");
    workloadName = benchmarkName + "_" + appName;
  }
  std::cout << "comp.ratio: " << mpctext::num(compStat->CompRatio) << std::endl;

  compStat->Print(workloadName, compOutputSavePath);
  compStat->PrintDetail(workloadName, compDetailedOutputSavePath);

  delete loader;
  delete compressor;
  return 0;
}

// The reference's per-line loop (main.cpp:208-248) as a batch loop: stream the file
// through the evaluator when the loader allows it, else pull batches, else lines.
static comp::CompResult *compressLines(comp::Compressor *compressor, trace::Loader *loader, bool perLine)
{
  if (compressor->GetLineSize() != loader->GetCachelineSize()) {
    printf("The trace has %u-byte lines but the evaluator is configured for %u-byte lines.\n",
           loader->GetCachelineSize(), compressor->GetLineSize());
    exit(1);
  }
  const std::string path = perLine ? std::string() : loader->GetStreamablePath();
  if (!path.empty()) {
    compressor->CompressFile(path);
    return compressor->GetResult();
  }
  if (!perLine && loader->SupportsBatch()) {
    const unsigned L = loader->GetCachelineSize();
    const unsigned long long cap = (64ull << 20) / L;
    std::vector<uint8_t> buf((size_t)(cap * L));
    for (;;) {
      unsigned long long n = loader->GetBatch(buf.data(), cap);
      if (n == 0) break;
      compressor->CompressBatch(buf.data(), n);
    }
  } else {
    // the reference's loop (main.cpp:208-248): one request object handed back and forth, a .log
    // trace's requests filtered to GLOBAL_ACC_R / GLOBAL_ACC_W (main.cpp:222-224)
    trace::gpgpusim::LoaderGPGPU *gpgpu = dynamic_cast<trace::gpgpusim::LoaderGPGPU *>(loader);
    trace::MemReq_t *memReq = gpgpu ? static_cast<trace::MemReq_t *>(new trace::gpgpusim::MemReqGPU_t)
                              : dynamic_cast<trace::apsim::LoaderGPGPU *>(loader) ? static_cast<trace::MemReq_t *>(new trace::apsim::MemReqGPU_t)
                                                                                  : new trace::MemReq_t;
    memReq->Reset();
    while (1) {
      memReq = loader->GetCacheline(memReq);
      if (memReq->isEnd) break;
      if (gpgpu) {
        const trace::gpgpusim::reqTypeGPU t = static_cast<trace::gpgpusim::MemReqGPU_t *>(memReq)->reqType;
        if (!(t == trace::gpgpusim::GLOBAL_ACC_R || t == trace::gpgpusim::GLOBAL_ACC_W)) continue;
      }
      compressor->CompressLine(memReq->data);
    }
    delete memReq;      // (the reference leaks its request object)
  }
  return compressor->GetResult();
}
