// FPC.h -- Frequent Pattern Compression baseline behind the reference's class names
// (reference src/compressor/FPC.h); evaluation on the MI355X via libmpc_hip.so.
#ifndef MPC_HOST_FPC_H
#define MPC_HOST_FPC_H

#include "CompResult.h"
#include "Compressor.h"

#define PREFIX_SIZE 3
#define NUM_FPC_PATTERN 8

namespace comp
{

enum class FPCState {
  Prefix0 = 0, Prefix1 = 1, Prefix2 = 2, Prefix3 = 3, Prefix4 = 4, Prefix5 = 5, Prefix6 = 6, Prefix7 = 7
};

struct FPCResult : public CompResult {
  FPCResult(unsigned lineSize) : CompResult(lineSize), Counts(NUM_FPC_PATTERN, 0), TotalWords(0) {}
  virtual void Update(unsigned uncompSize, unsigned compSize, int selected)
  {
    CompResult::Update(uncompSize, compSize);
    TotalWords++;
    Counts[(size_t)selected]++;
  }
  virtual void Print(std::string workloadName = "", std::string filePath = "");
  void LoadVector(const uint64_t *vec);
  std::vector<uint64_t> Counts;
  uint64_t TotalWords;
};

class FPC : public Compressor
{
public:
  FPC(unsigned lineSize);
  virtual ~FPC();
  virtual unsigned CompressLine(std::vector<uint8_t> &dataLine);
  virtual CompResult *GetResult();
  virtual void CompressBatch(const uint8_t *lines, unsigned long long n);
  virtual unsigned long long CompressFile(const std::string &tracePath);
  virtual unsigned GetLineSize() { return m_LineSize; }

private:
  mpc_handle *m_Handle;
  unsigned m_LineSize;
};

}  // namespace comp

#endif
