// BDI.h -- Base-Delta-Immediate baseline behind the reference's class names
// (reference src/compressor/BDI.h); evaluation on the MI355X via libmpc_hip.so.
#ifndef MPC_HOST_BDI_H
#define MPC_HOST_BDI_H

#include "CompResult.h"
#include "Compressor.h"

namespace comp
{

enum class BDIState {
  Zeros = 0, Repeat = 1, Base8Delta1 = 2, Base8Delta2 = 3, Base8Delta4 = 4,
  Base4Delta1 = 5, Base4Delta2 = 6, Base2Delta1 = 7, Uncompressed = 8
};

struct BDIResult : public CompResult {
  BDIResult(unsigned lineSize) : CompResult(lineSize), Counts(9, 0) {}
  virtual void Update(unsigned uncompSize, unsigned compSize, int selected)
  {
    CompResult::Update(uncompSize, compSize);
    Counts[(size_t)selected]++;
  }
  virtual void Print(std::string workloadName = "", std::string filePath = "");
  void LoadVector(const uint64_t *vec);
  std::vector<uint64_t> Counts;
};

class BDI : public Compressor
{
public:
  BDI(unsigned lineSize);
  virtual ~BDI();
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
