// Compressor.h -- evaluator interface, source-compatible with the reference's
// comp::Compressor (reference src/compressor/Compressor.h:18-33).  ADDITIVE:
// CompressBatch()/CompressFile(), which the driver uses when the loader can hand
// out many lines at once; the per-line CompressLine() keeps its signature and
// meaning (it evaluates one line on the device and returns its size in bits).
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

  // ADDITIVE: n consecutive lines of GetLineSize() bytes; statistics accumulate.
  virtual void CompressBatch(const uint8_t *lines, unsigned long long n) = 0;
  // ADDITIVE: stream a whole trace file through the evaluator: a C-order uint8 [N, L] .npy
  // file (all rows but the last, as the reference driver does) or a GPGPU-Sim .log file (its
  // GLOBAL_ACC_R / GLOBAL_ACC_W requests); returns the number of lines evaluated.
  virtual unsigned long long CompressFile(const std::string &tracePath) = 0;
  virtual unsigned GetLineSize() = 0;

protected:
  CompResult *m_Stat;
};

}  // namespace comp

#endif
