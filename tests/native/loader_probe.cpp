// loader_probe.cpp -- test driver for the host-side mirrors of trace::LoaderNPY and
// trace::gpgpusim::LoaderGPGPU (CPU only).  Prints what the per-line interface
// (GetCacheline, as the reference driver uses it, src/main.cpp:208-248, including its
// request-type filter for .log traces) and the additive batch interface (GetBatch) deliver.
//   loader_probe FILE.npy|FILE.log|FILE.txt line|batch [batch_lines] [apsim_line_size]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "LoaderAPSim.h"
#include "LoaderGPGPU.h"
#include "LoaderNPY.h"

static unsigned long long fnv(const uint8_t *p, size_t n, unsigned long long h)
{
  for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

int main(int argc, char **argv)
{
  if (argc < 3) return 2;
  const std::string path = argv[1];
  const bool is_log = path.size() > 4 && path.compare(path.size() - 4, 4, ".log") == 0;
  const bool is_txt = path.size() > 4 && path.compare(path.size() - 4, 4, ".txt") == 0;
  const unsigned txt_line = argc > 4 ? (unsigned)strtoul(argv[4], nullptr, 10) : 32u;
  trace::Loader *lp = is_log ? static_cast<trace::Loader *>(new trace::gpgpusim::LoaderGPGPU(path))
                      : is_txt ? static_cast<trace::Loader *>(new trace::apsim::LoaderGPGPU(path, txt_line))
                               : static_cast<trace::Loader *>(new trace::LoaderNPY(path));
  trace::Loader &loader = *lp;
  const unsigned L = loader.GetCachelineSize();
  printf("lines %llu line_size %u\n", loader.GetNumLines(), L);
  unsigned long long h = 1469598103934665603ull, n = 0;
  if (!strcmp(argv[2], "line")) {
    trace::MemReq_t *req = is_log ? static_cast<trace::MemReq_t *>(new trace::gpgpusim::MemReqGPU_t)
                           : is_txt ? static_cast<trace::MemReq_t *>(new trace::apsim::MemReqGPU_t) : new trace::MemReq_t;
    req->Reset();
    while (1) {
      req = loader.GetCacheline(req);
      if (req->isEnd) break;                 // the reference tests isEnd BEFORE compressing
      if (is_log && !trace::gpgpusim::LoaderGPGPU::isEvaluated(
                        (uint32_t) static_cast<trace::gpgpusim::MemReqGPU_t *>(req)->reqType))
        continue;                            // main.cpp:222-224
      if (req->data.size() != L || (!is_txt && req->reqSize != L)) return 3;   // (the APSim loader reports reqSize 64 for 32-byte beats, as the reference does)
      h = fnv(req->data.data(), L, h);
      n++;
    }
    delete req;   // (the reference driver leaks it; the sanitizer build must not)
  } else {
    const unsigned long long cap = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1000;
    std::vector<uint8_t> buf(cap * L);
    for (;;) {
      unsigned long long got = loader.GetBatch(buf.data(), cap);
      if (!got) break;
      h = fnv(buf.data(), got * L, h);
      n += got;
    }
  }
  printf("delivered %llu hash %llu\n", n, h);
  delete lp;
  return 0;
}
