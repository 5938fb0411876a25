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

struct CompResult {
  CompResult(unsigned lineSize) : LineSize(lineSize), OriginalSize(0), CompressedSize(0), CompRatio(0) {}
  virtual ~CompResult() {}

  virtual void Update(unsigned uncompSize, unsigned compSize, int selected = 0)
  {
    (void)selected;
    OriginalSize += uncompSize;
    CompressedSize += compSize;
    CompRatio = (double)OriginalSize / (double)CompressedSize;
  }

  // opens filePath for appending, writing `header` first if the file is new
  static void openForAppend(std::ofstream &file, const std::string &filePath, const std::string &header)
  {
    if (!isFileExists(filePath)) {
      file.open(filePath);
      if (!file.is_open()) {
        std::cout << "File is not open: \"" << filePath << "\"" << std::endl;
        exit(1);
      }
      file << header;
      file.close();
    }
    file.open(filePath, std::ios_base::app);
  }

  virtual void Print(std::string workloadName = "", std::string filePath = "")
  {
    std::ofstream file;
    if (filePath != "") openForAppend(file, filePath, "workload,original_size,compressed_size,compression_ratio,\n");
    std::ostream &stream = (filePath == "") ? std::cout : file;
    stream << workloadName << "," << OriginalSize << "," << CompressedSize << "," << mpctext::num(CompRatio) << ","
           << std::endl;
  }

  virtual void PrintDetail(std::string workloadName = "", std::string filePath = "")
  {
    (void)workloadName;
    (void)filePath;
  }

  std::string CompressorName;
  const unsigned LineSize;
  uint64_t OriginalSize;
  uint64_t CompressedSize;
  double CompRatio;
};

}  // namespace comp

#endif
