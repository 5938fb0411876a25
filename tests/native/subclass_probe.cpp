// subclass_probe.cpp -- a compressor written against the REFERENCE's two-method interface (comp::Compressor with
// CompressLine() and GetResult() only, reference src/compressor/Compressor.h:18-33, in the style of its CPACK / SC2
// classes) must compile and run against cal_22-mpc_amd/host/Compressor.h unchanged: the additive CompressBatch() /
// CompressFile() / GetLineSize() are defaulted.  CPU only (no device library is linked).  Test infrastructure.
//
//   subclass_probe TRACE.npy|TRACE.log LINE_SIZE  ->  "line <n> <orig> <comp>", "batch ...", "file ..." (equal)
#include <cstdio>
#include <string>
#include <vector>

#include "Compressor.h"
#include "LoaderGPGPU.h"
#include "LoaderNPY.h"
#include "utils.h"

namespace comp
{
// "compressed size" = 8 bits per non-zero byte + 1: enough to tell the routes apart from one another
class NonZeroBytes : public Compressor
{
public:
  explicit NonZeroBytes(unsigned lineSize)
  {
    m_Stat = new CompResult(lineSize);
    m_Stat->CompressorName = "NonZeroBytes";
  }
  ~NonZeroBytes() { delete m_Stat; }      // (the reference's classes leak it; the sanitizer build here would say so)
  virtual unsigned CompressLine(std::vector<uint8_t> &dataLine)
  {
    unsigned bits = 1;
    for (uint8_t b : dataLine) bits += b ? 8 : 0;
    m_Stat->Update(dataLine.size() * 8, bits);
    return bits;
  }
};
}  // namespace comp

static void report(const char *tag, unsigned long long n, comp::Compressor *c)
{
  comp::CompResult *r = c->GetResult();
  std::printf("%s %llu %llu %llu\n", tag, n, (unsigned long long)r->OriginalSize, (unsigned long long)r->CompressedSize);
}

int main(int argc, char **argv)
{
  if (argc != 3) return 2;
  const std::string path = argv[1];
  const unsigned L = (unsigned)std::atoi(argv[2]);
  const bool log = mpctext::ends_with(path, ".log");
  // (a) the reference driver's loop, one line per call
  {
    comp::Compressor *c = new comp::NonZeroBytes(L);
    trace::Loader *loader = log ? static_cast<trace::Loader *>(new trace::gpgpusim::LoaderGPGPU(path))
                                : static_cast<trace::Loader *>(new trace::LoaderNPY(path));
    trace::MemReq_t *req = log ? static_cast<trace::MemReq_t *>(new trace::gpgpusim::MemReqGPU_t) : new trace::MemReq_t;
    unsigned long long n = 0;
    req->Reset();
    while (1) {
      req = loader->GetCacheline(req);
      if (req->isEnd) break;
      if (log) {
        const trace::gpgpusim::reqTypeGPU t = static_cast<trace::gpgpusim::MemReqGPU_t *>(req)->reqType;
        if (!(t == trace::gpgpusim::GLOBAL_ACC_R || t == trace::gpgpusim::GLOBAL_ACC_W)) continue;
      }
      c->CompressLine(req->data);
      n++;
    }
    report("line", n, c);
    delete req;
    delete loader;
    delete c;
  }
  // (b) the defaulted CompressBatch() over the loader's batches
  {
    comp::Compressor *c = new comp::NonZeroBytes(L);
    trace::Loader *loader = log ? static_cast<trace::Loader *>(new trace::gpgpusim::LoaderGPGPU(path))
                                : static_cast<trace::Loader *>(new trace::LoaderNPY(path));
    std::vector<uint8_t> buf((size_t)L * 100);
    unsigned long long n = 0;
    for (;;) {
      const unsigned long long got = loader->GetBatch(buf.data(), 100);
      if (got == 0) break;
      c->CompressBatch(buf.data(), got);
      n += got;
    }
    report("batch", n, c);
    delete loader;
    delete c;
  }
  // (c) the defaulted CompressFile() and GetLineSize()
  {
    comp::Compressor *c = new comp::NonZeroBytes(L);
    if (c->GetLineSize() != L) return 3;
    const unsigned long long n = c->CompressFile(path);
    report("file", n, c);
    delete c;
  }
  return 0;
}
