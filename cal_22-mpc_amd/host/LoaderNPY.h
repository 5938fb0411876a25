// LoaderNPY.h -- C-order uint8 [N, L] .npy traces (reference
// src/loader/LoaderNPY.{h,cpp}).  The reference reads the file through the
// un-vendored libnpy; this one parses the published NPY container itself
// (magic "\x93NUMPY", version, header length, dict with descr / fortran_order /
// shape).  Behaviour kept: the whole array is resident, rows are served in
// order, and isEnd is raised ON the last row, so a driver that tests isEnd
// before compressing never sees that row (LoaderNPY.cpp:28-32 + main.cpp:240).
#ifndef MPC_HOST_LOADERNPY_H
#define MPC_HOST_LOADERNPY_H

#include "Loader.h"

namespace trace
{

class LoaderNPY : public Loader
{
public:
  LoaderNPY(const char *filePath);
  LoaderNPY(const std::string filePath);

  virtual MemReq_t *GetCacheline(MemReq_t *memReq);
  virtual unsigned GetCachelineSize();
  virtual unsigned long long GetNumLines();
  virtual void Reset();

  virtual unsigned long long GetBatch(uint8_t *dst, unsigned long long maxLines);
  virtual bool SupportsBatch() { return true; }
  virtual std::string GetStreamablePath();

private:
  void readHeader();

  std::vector<WORD_SIZE> m_DataLines;   // filled lazily: only the per-line interface needs it
  std::vector<uint64_t> m_DataShape;
  uint64_t m_CurrentLine;
  uint64_t m_DataOffset;
  bool m_Loaded;
};

}  // namespace trace

#endif
