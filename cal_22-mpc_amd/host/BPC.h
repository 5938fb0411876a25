// BPC.h -- Bit-Plane Compression baseline behind the reference's class names
// (reference src/compressor/BPC.h); evaluation on the MI355X via libmpc_hip.so.
#ifndef MPC_HOST_BPC_H
#define MPC_HOST_BPC_H

#include "CompResult.h"
#include "Compressor.h"

#define NUM_BPC_PATTERN 7

namespace comp
{

enum class BPCPattern { Uncomp = 0, ZRLE = 1, Zero = 2, SingleOne = 3, ConsecTwoOnes = 4, ZeroDBP = 5, AllOnes = 6 };

struct BPCResult : public CompResult {
  BPCResult(unsigned lineSize) : CompResult(lineSize), Counts(NUM_BPC_PATTERN, 0), TotalWords(0) {}
  void UpdatePattern(unsigned numWords, int selected)
  {
    TotalWords += numWords;
    Counts[(size_t)selected]++;
  }
  virtual void Print(std::string workloadName = "", std::string filePath = "");
  void LoadVector(const uint64_t *vec);
  std::vector<uint64_t> Counts;
  uint64_t TotalWords;
};

class BPC : public Compressor
{
public:
  BPC(unsigned lineSize);
  virtual ~BPC();
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
