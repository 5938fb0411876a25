// Compressor.cpp -- default of the additive comp::Compressor::CompressFile(): the reference driver's own loop
// (src/main.cpp:208-248) over a loader chosen by the file's extension (main.cpp:74-84), for evaluators that only
// implement CompressLine().  The GPU evaluators override it with a streamed batch path.
#include "Compressor.h"

#include <cstdio>
#include <cstdlib>

#include "LoaderAPSim.h"
#include "LoaderGPGPU.h"
#include "LoaderNPY.h"
#include "utils.h"

namespace comp
{

unsigned long long Compressor::CompressFile(const std::string &tracePath)
{
  trace::Loader *loader = nullptr;
  trace::MemReq_t *memReq = nullptr;
  trace::gpgpusim::LoaderGPGPU *gpgpu = nullptr;
  if (mpctext::ends_with(tracePath, ".npy")) {
    loader = new trace::LoaderNPY(tracePath);
    memReq = new trace::MemReq_t;
  } else if (mpctext::ends_with(tracePath, ".log")) {
    loader = gpgpu = new trace::gpgpusim::LoaderGPGPU(tracePath);
    memReq = new trace::gpgpusim::MemReqGPU_t;
  } else if (mpctext::ends_with(tracePath, ".txt")) {
    loader = new trace::apsim::LoaderGPGPU(tracePath, GetLineSize());
    memReq = new trace::apsim::MemReqGPU_t;
  } else {
    printf("Invalid File! %s\n", tracePath.c_str());
    exit(1);
  }
  unsigned long long done = 0;
  memReq->Reset();
  while (1) {
    memReq = loader->GetCacheline(memReq);
    if (memReq->isEnd) break;
    if (gpgpu) {     // main.cpp:222-224
      const trace::gpgpusim::reqTypeGPU t = static_cast<trace::gpgpusim::MemReqGPU_t *>(memReq)->reqType;
      if (!(t == trace::gpgpusim::GLOBAL_ACC_R || t == trace::gpgpusim::GLOBAL_ACC_W)) continue;
    }
    CompressLine(memReq->data);
    done++;
  }
  delete memReq;
  delete loader;
  return done;
}

}  // namespace comp
