// perline_probe.cpp -- drives comp::VPC / comp::BDI (cal_22-mpc_amd/host) exactly as the reference's
// driver does (src/main.cpp:208-248): one MemReq_t handed back and forth,
//     GetCacheline -> isEnd? -> CompressLine(memReq->data) -> ... -> GetResult -> Print / PrintDetail
// i.e. the UNCHANGED per-line interface of a reference driver, one line per call.  Writes the CSV rows
// through the result object's own Print / PrintDetail, every returned size as uint16 to SIZES.bin, and
// prints the line rate.  Test infrastructure (tests/test_cli.py), not product code.
//
//   perline_probe VPC cfg.json trace.npy out_results.csv out_detail.csv sizes.bin [LINE_BUFFER]
//   perline_probe BDI -        trace.npy out_results.csv out_detail.csv sizes.bin [LINE_BUFFER]
// LINE_BUFFER > 0: the same loop after Compressor::SetLineBuffering(LINE_BUFFER) (CompressLine returns 0 then).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "BDI.h"
#include "LoaderNPY.h"
#include "VPC.h"

int main(int argc, char **argv)
{
  if (argc != 7 && argc != 8) {
    std::fprintf(stderr, "usage: perline_probe VPC|BDI CONFIG|- TRACE.npy RESULTS.csv DETAIL.csv SIZES.bin [LINE_BUFFER]\n");
    return 2;
  }
  const std::string algo = argv[1];
  trace::Loader *loader = new trace::LoaderNPY(std::string(argv[3]));
  comp::Compressor *compressor = nullptr;
  if (algo == "VPC") compressor = new comp::VPC(std::string(argv[2]));
  else compressor = new comp::BDI(loader->GetCachelineSize());

  if (argc == 8) compressor->SetLineBuffering(std::strtoull(argv[7], nullptr, 10));

  std::vector<uint16_t> sizes;
  trace::MemReq_t *memReq = new trace::MemReq_t;
  memReq->Reset();
  const auto t0 = std::chrono::steady_clock::now();
  while (1) {
    memReq = loader->GetCacheline(memReq);
    if (memReq->isEnd) break;
    std::vector<uint8_t> &dataLine = memReq->data;
    sizes.push_back((uint16_t)compressor->CompressLine(dataLine));
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  comp::CompResult *stat = compressor->GetResult();
  stat->Print("probe_trace", argv[4]);
  stat->PrintDetail("probe_trace", argv[5]);
  FILE *f = std::fopen(argv[6], "wb");
  if (!f) return 3;
  std::fwrite(sizes.data(), sizeof(uint16_t), sizes.size(), f);
  std::fclose(f);
  std::printf("%s per-line: %zu lines in %.3f s = %.0f lines/s, comp.ratio %s\n", algo.c_str(), sizes.size(), dt,
              sizes.size() / dt, mpctext::num(stat->CompRatio).c_str());
  delete loader;
  delete compressor;
  return 0;
}
