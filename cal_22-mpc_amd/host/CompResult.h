// CompResult.h -- statistics base class, source-compatible with the reference's
// comp::CompResult (reference src/compressor/CompResult.h:24-86): same members,
// same Update() arithmetic, same CSV text.
#ifndef MPC_HOST_COMPRESULT_H
#define MPC_HOST_COMPRESULT_H

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "Loader.h"
#include "utils.h"

#define BYTE (8)
#define BYTEMAX (0xff)
#define BYTE2MAX (0xffff)
#define BYTE4MAX (0xffffffff)
#define BYTE8MAX (0xffffffffffffffff)

#define COMPSIZELIMIT ((ACCESS_GRAN * BYTE) + 32)

namespace comp
{

// Totals every evaluator keeps (reference CompResult.h:24-86).  The text output lives in
// CompResult.cpp.
struct CompResult {
  explicit CompResult(unsigned lineSize) : LineSize(lineSize) {}
  virtual ~CompResult() {}

  // one evaluated line (or word): running totals and ratio
  virtual void Update(unsigned uncompSize, unsigned compSize, int selected = 0);

  // "workload,original_size,compressed_size,compression_ratio," row, to stdout or appended to filePath
  virtual void Print(std::string workloadName = "", std::string filePath = "");
  virtual void PrintDetail(std::string workloadName = "", std::string filePath = "");

  // opens filePath for appending, writing `header` first if the file is new
  static void openForAppend(std::ofstream &file, const std::string &filePath, const std::string &header);

  std::string CompressorName;
  const unsigned LineSize;
  uint64_t OriginalSize = 0;
  uint64_t CompressedSize = 0;
  double CompRatio = 0;
};

}  // namespace comp

#endif
