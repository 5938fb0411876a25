// VPC.h -- the multi-prediction evaluator behind the reference's class names
// (reference src/compressor/VPC.h).  comp::VPC(configPath) reads the same JSON
// configuration; CompressLine() returns the same bit count; VPCResult holds the
// same statistics and writes the same two CSV files.  The evaluation itself runs
// on the MI355X through libmpc_hip.so (include/mpc_hip.h) -- there is no CPU
// evaluation in this class.
#ifndef MPC_HOST_VPC_H
#define MPC_HOST_VPC_H

#include <map>
#include <utility>

#include "CompResult.h"
#include "Compressor.h"

namespace comp
{

struct ClusterStat {
  ClusterStat() : count(0), originalSize(0), compressedSize(0), compRatio(0)
  {
    for (int i = 0; i < COMPSIZELIMIT; i++) compSizeHistogram.insert(std::make_pair(i, 0));
  }
  uint64_t count;
  uint64_t originalSize;
  uint64_t compressedSize;
  double compRatio;
  std::map<int, uint64_t> compSizeHistogram;
};

struct VPCResult : public CompResult {
  VPCResult(unsigned lineSize) : CompResult(lineSize), m_NumModules(0) {}
  VPCResult(unsigned lineSize, int numModules) : CompResult(lineSize), m_NumModules(numModules) { SetNumModules(numModules); }

  virtual void Update(unsigned uncompSize, unsigned compSize, int selected);
  void UpdateResidueStat(double mae, double mse, const int chosenCompModule);
  virtual void Print(std::string workloadName = "", std::string filePath = "");
  virtual void PrintDetail(std::string workloadName = "", std::string filePath = "");
  void SetNumModules(int numModules);

  // Replace every statistic by the integer vector of mpc_stats_get() (layout in
  // include/mpc_hip.h).  Doubles are derived exactly as the reference derives
  // them: ratio = (double)orig / (double)comp; MAE = (sum_r / L) / lines.
  void LoadVector(const uint64_t *vec, int numModules, int histBins);

  std::map<int, ClusterStat> m_ClusterStats;
  std::map<int, double> m_SumMAE, m_MAE;
  std::map<int, double> m_SumMSE, m_MSE;
  std::map<int, uint64_t> m_NumLines;
  int m_NumModules;
};

class VPC : public Compressor
{
public:
  VPC(std::string configPath);
  virtual ~VPC();

  int GetCachelineSize() { return m_LineSize; }
  int GetNumModules() { return m_NumModules; }
  int GetNumClusters() { return m_NumClusters; }

  virtual unsigned CompressLine(std::vector<uint8_t> &dataLine);
  virtual CompResult *GetResult();
  virtual void CompressBatch(const uint8_t *lines, unsigned long long n);
  virtual unsigned long long CompressFile(const std::string &tracePath);
  virtual unsigned GetLineSize() { return (unsigned)m_LineSize; }

private:
  mpc_handle *m_Handle;
  int m_LineSize;
  int m_NumModules;
  int m_NumClusters;
  int m_HistBins;
};

}  // namespace comp

#endif
