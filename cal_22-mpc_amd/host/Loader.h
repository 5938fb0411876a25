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

// request direction (reference Loader.h:17-22)
enum rw_t { READ, WRITE, NA };

// One memory request / cache line as the loaders hand it out (reference Loader.h:24-60).  The
// driver allocates one object and passes it to every GetCacheline() call; the loader refills it.
struct MemReq_t {
  addr_t addr = 0;
  rw_t rw = NA;
  uint32_t reqSize = 0;            // payload bytes
  std::vector<WORD_SIZE> data;     // the line
  bool isEnd = false;              // raised by the read that runs off the trace

  virtual ~MemReq_t() {}

  // back to the empty request
  virtual void Reset()
  {
    data.clear();
    addr = 0, reqSize = 0;
    rw = NA;
    isEnd = false;
  }

  // the reference copies through a non-const reference; both spellings are kept
  MemReq_t &operator=(MemReq_t &other)
  {
    if (this != &other) {
      data = other.data;
      addr = other.addr, reqSize = other.reqSize;
      rw = other.rw;
      isEnd = other.isEnd;
    }
    return *this;
  }
  void Set(MemReq_t &other) { operator=(other); }
};

// Base class of the trace readers (reference Loader.h:62-88): opens the file in binary mode,
// serves lines one at a time.
class Loader
{
public:
  explicit Loader(const char *filePath) : m_FilePath(filePath ? filePath : "") { openStream(); }
  explicit Loader(const std::string filePath) : m_FilePath(filePath) { openStream(); }
  virtual ~Loader() {}

  virtual MemReq_t *GetCacheline(MemReq_t *) = 0;      // fills and returns its argument
  virtual unsigned GetCachelineSize() = 0;             // bytes per line
  virtual unsigned long long GetNumLines() = 0;
  virtual void Reset() = 0;                            // rewind

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

private:
  void openStream() { m_FileStream.open(m_FilePath.c_str(), std::ios_base::in | std::ios_base::binary); }
};

}  // namespace trace

#endif
