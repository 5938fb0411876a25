// Compressor.h -- evaluator interface, source-compatible with the reference's
// comp::Compressor (reference src/compressor/Compressor.h:18-33): a class that overrides only
// CompressLine() (and GetResult()), like the reference's CPACK / SC2 / ... compressors, compiles
// and runs against this header unchanged.  ADDITIVE and DEFAULTED: CompressBatch() / CompressFile() /
// GetLineSize(), which the driver uses when the loader can hand out many lines at once -- the
// defaults fall back to CompressLine() per line; the GPU evaluators (VPC, BDI, FPC, BPC) override
// them.  The per-line CompressLine() keeps its signature and meaning (it evaluates one line and
// returns its size in bits), unless the driver opts into SetLineBuffering().
#ifndef MPC_HOST_COMPRESSOR_H
#define MPC_HOST_COMPRESSOR_H

#include <fstream>
#include <ios>
#include <iostream>
#include <string>
#include <vector>

#include "CompResult.h"
#include "Loader.h"

struct mpc_handle;

namespace comp
{

class Compressor
{
public:
  virtual ~Compressor() {}
  std::string GetCompressorName() { return m_Stat->CompressorName; }

  virtual unsigned CompressLine(std::vector<uint8_t> &dataLine) = 0;
  // Refreshes m_Stat from the device-side statistics and returns it (borrowed).
  virtual CompResult *GetResult() { return m_Stat; }

  // ADDITIVE: n consecutive lines of GetLineSize() bytes; statistics accumulate.  Default: CompressLine() per line.
  virtual void CompressBatch(const uint8_t *lines, unsigned long long n)
  {
    const unsigned L = GetLineSize();
    std::vector<uint8_t> line(L);
    for (unsigned long long i = 0; i < n; i++) {
      line.assign(lines + i * L, lines + (i + 1) * L);
      CompressLine(line);
    }
  }
  // ADDITIVE: stream a whole trace file through the evaluator: a C-order uint8 [N, L] .npy
  // file (all rows but the last, as the reference driver does) or a GPGPU-Sim .log file (its
  // GLOBAL_ACC_R / GLOBAL_ACC_W requests); returns the number of lines evaluated.  Default
  // (Compressor.cpp): the reference driver's loop, GetCacheline() -> CompressLine() per line.
  virtual unsigned long long CompressFile(const std::string &tracePath);
  // ADDITIVE: bytes per line.  Default: the result object's LineSize (CompResult.h).
  virtual unsigned GetLineSize() { return m_Stat->LineSize; }

  // ADDITIVE: buffered per-line mode, for drivers that keep the reference's loop (GetCacheline ->
  // CompressLine per line, main.cpp:225-243 -- which never looks at CompressLine()'s return value).
  // After SetLineBuffering(n), n > 0, CompressLine() copies the line into a host buffer of n lines and
  // RETURNS 0; the buffer is evaluated as one batch when it is full, and by GetResult(), CompressBatch()
  // and CompressFile().  The statistics are the same as without buffering.  Off (0) by default: every
  // CompressLine() call then evaluates its line on the device and returns its size in bits.
  void SetLineBuffering(unsigned long long lines)
  {
    FlushLines();
    m_BufCap = lines;
    m_LineBuf.clear();
    m_LineBuf.shrink_to_fit();
  }

protected:
  bool LineBuffering() const { return m_BufCap != 0; }
  void BufferLine(const std::vector<uint8_t> &dataLine)
  {
    if (m_LineBuf.capacity() == 0) m_LineBuf.reserve((size_t)(m_BufCap * dataLine.size()));
    m_LineBuf.insert(m_LineBuf.end(), dataLine.begin(), dataLine.end());
    if (++m_BufLines == m_BufCap) FlushLines();
  }
  void FlushLines()
  {
    if (m_BufLines == 0) return;
    const unsigned long long n = m_BufLines;
    m_BufLines = 0;                 // (CompressBatch() flushes first: nothing left to flush then)
    CompressBatch(m_LineBuf.data(), n);
    m_LineBuf.clear();
  }

  CompResult *m_Stat;

private:
  std::vector<uint8_t> m_LineBuf;
  unsigned long long m_BufLines = 0, m_BufCap = 0;
};

}  // namespace comp

#endif
