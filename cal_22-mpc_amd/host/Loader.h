// Loader.h -- trace loader interface, kept source-compatible with the reference's
// trace::Loader / trace::MemReq_t (reference src/loader/Loader.h:17-88) so code
// written against it keeps compiling.  ADDITIVE: GetBatch(), which hands out many
// lines at once for the batch evaluator (the per-line GetCacheline() still works).
#ifndef MPC_HOST_LOADER_H
#define MPC_HOST_LOADER_H

#include <cstdint>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#define WORD_SIZE uint8_t
#define ACCESS_GRAN 32

typedef uint64_t addr_t;

namespace trace
{

enum rw_t { READ, WRITE, NA };

struct MemReq_t {
  addr_t addr;
  rw_t rw;
  uint32_t reqSize;
  std::vector<WORD_SIZE> data;
  bool isEnd;

  virtual ~MemReq_t() {}
  virtual void Reset()
  {
    addr = 0;
    rw = NA;
    reqSize = 0;
    data.clear();
    isEnd = false;
  }
  void Set(MemReq_t &memReq) { *this = memReq; }
  MemReq_t &operator=(MemReq_t &rhs)
  {
    addr = rhs.addr;
    rw = rhs.rw;
    reqSize = rhs.reqSize;
    data = rhs.data;
    isEnd = rhs.isEnd;
    return *this;
  }
};

class Loader
{
public:
  Loader(const char *filePath) : m_FilePath(filePath)
  {
    m_FileStream.open(m_FilePath, std::ios_base::in | std::ios_base::binary);
  }
  Loader(const std::string filePath) : m_FilePath(filePath)
  {
    m_FileStream.open(filePath.c_str(), std::ios_base::in | std::ios_base::binary);
  }
  virtual ~Loader() {}

  virtual MemReq_t *GetCacheline(MemReq_t *) = 0;
  virtual unsigned GetCachelineSize() = 0;
  virtual unsigned long long GetNumLines() = 0;
  virtual void Reset() = 0;

  // ADDITIVE.  Copies up to maxLines of the lines the per-line interface would
  // still deliver (i.e. honouring the loader's isEnd convention) into dst and
  // returns how many were copied; 0 when the trace is exhausted.  Loaders that
  // do not implement it return 0 and the driver falls back to GetCacheline().
  virtual unsigned long long GetBatch(uint8_t *dst, unsigned long long maxLines)
  {
    (void)dst;
    (void)maxLines;
    return 0;
  }
  // ADDITIVE.  true when GetBatch() is implemented (an exhausted or empty trace then simply yields 0)
  virtual bool SupportsBatch() { return false; }
  // ADDITIVE.  Path of a file the evaluator may stream itself (mpc_compress_npy),
  // or "" if the loader has to be read through GetBatch()/GetCacheline().
  virtual std::string GetStreamablePath() { return ""; }

protected:
  const std::string m_FilePath;
  std::ifstream m_FileStream;
};

}  // namespace trace

#endif
