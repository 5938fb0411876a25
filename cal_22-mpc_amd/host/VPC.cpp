#include "VPC.h"

#include <cstdio>
#include <cstdlib>

#include "mpc_hip.h"

namespace comp
{

// ---------------------------------------------------------------------------
// VPCResult (reference VPC.h:36-238)
// ---------------------------------------------------------------------------
void VPCResult::SetNumModules(int numModules)
{
  m_NumModules = numModules;
  for (int i = -1; i < numModules; i++) {
    m_ClusterStats.insert(std::make_pair(i, ClusterStat()));
    m_SumMAE.insert(std::make_pair(i, 0));
    m_MAE.insert(std::make_pair(i, 0));
    m_SumMSE.insert(std::make_pair(i, 0));
    m_MSE.insert(std::make_pair(i, 0));
    m_NumLines.insert(std::make_pair(i, 0));
  }
}

void VPCResult::Update(unsigned uncompSize, unsigned compSize, int selected)
{
  CompResult::Update(uncompSize, compSize);
  ClusterStat &c = m_ClusterStats[selected];
  c.originalSize += uncompSize;
  c.compressedSize += compSize;
  c.compRatio = (double)c.originalSize / (double)c.compressedSize;
  c.count++;
  c.compSizeHistogram[(int)compSize]++;
}

void VPCResult::UpdateResidueStat(double mae, double mse, const int chosenCompModule)
{
  m_SumMAE[chosenCompModule] += mae;
  m_SumMSE[chosenCompModule] += mse;
  uint64_t &n = m_NumLines[chosenCompModule];
  n++;
  m_MAE[chosenCompModule] = m_SumMAE[chosenCompModule] / (double)n;
  m_MSE[chosenCompModule] = m_SumMSE[chosenCompModule] / (double)n;
}

void VPCResult::LoadVector(const uint64_t *v, int numModules, int histBins)
{
  const int K = numModules + 1;
  m_ClusterStats.clear();
  m_SumMAE.clear(); m_MAE.clear(); m_SumMSE.clear(); m_MSE.clear(); m_NumLines.clear();
  SetNumModules(numModules);
  OriginalSize = v[1];
  CompressedSize = v[2];
  CompRatio = v[0] ? (double)OriginalSize / (double)CompressedSize : 0.0;
  for (int k = 0; k < K; k++) {
    const int cluster = k - 1;
    ClusterStat &c = m_ClusterStats[cluster];
    c.count = v[3 + 6 * k + 0];
    c.originalSize = v[3 + 6 * k + 1];
    c.compressedSize = v[3 + 6 * k + 2];
    c.compRatio = c.count ? (double)c.originalSize / (double)c.compressedSize : 0.0;
    const uint64_t lines = v[3 + 6 * k + 3];
    m_NumLines[cluster] = lines;
    // per line the reference adds (sum over bytes) / L to a running double
    m_SumMAE[cluster] = (double)v[3 + 6 * k + 4] / (double)LineSize;
    m_SumMSE[cluster] = (double)v[3 + 6 * k + 5] / (double)LineSize;
    m_MAE[cluster] = lines ? m_SumMAE[cluster] / (double)lines : 0.0;
    m_MSE[cluster] = lines ? m_SumMSE[cluster] / (double)lines : 0.0;
    for (int s = 0; s < histBins; s++) {
      const uint64_t n = v[3 + 6 * K + (uint64_t)k * histBins + s];
      if (n || s < COMPSIZELIMIT) c.compSizeHistogram[s] = n;
    }
  }
}

void VPCResult::Print(std::string workloadName, std::string filePath)
{
  std::ofstream file;
  if (filePath != "") {
    std::string header = "workload,total,,,";
    for (int i = -1; i < m_NumModules; i++) header += std::to_string(i) + ",,,";
    header += "\n,original_size,compressed_size,compression_ratio,count,";
    for (int i = -1; i < m_NumModules; i++) header += "original_size,compressed_size,compression_ratio,";
    header += "\n";
    openForAppend(file, filePath, header);
  }
  std::ostream &stream = (filePath == "") ? std::cout : file;
  stream << workloadName << "," << OriginalSize << "," << CompressedSize << "," << mpctext::num(CompRatio) << ",";
  for (int i = -1; i < m_NumModules; i++) {
    ClusterStat &c = m_ClusterStats[i];
    stream << c.originalSize << "," << c.compressedSize << "," << mpctext::num(c.compRatio) << ",";
  }
  stream << std::endl;
}

void VPCResult::PrintDetail(std::string workloadName, std::string filePath)
{
  std::ofstream file;
  if (filePath != "") {
    std::string header = "workload,";
    for (int i = -1; i < m_NumModules; i++) header += std::to_string(i) + ",,";
    for (int i = 0; i < m_NumModules; i++) {
      header += std::to_string(i) + ",";
      for (int j = 0; j < COMPSIZELIMIT - 1; j++) header += ",";
    }
    header += "\n,";
    for (int i = -1; i < m_NumModules; i++) header += "mae,mse,";
    for (int i = 0; i < m_NumModules; i++)
      for (int j = 0; j < COMPSIZELIMIT; j++) header += std::to_string(j) + ",";
    header += "\n";
    openForAppend(file, filePath, header);
  }
  std::ostream &stream = (filePath == "") ? std::cout : file;
  stream << workloadName << ",";
  for (int i = -1; i < m_NumModules; i++) stream << mpctext::num(m_MAE[i]) << "," << mpctext::num(m_MSE[i]) << ",";
  // histogram of clusters 0..M-1 only, bins 0..COMPSIZELIMIT-1 only (as the reference prints)
  for (int i = 0; i < m_NumModules; i++) {
    ClusterStat &c = m_ClusterStats[i];
    for (int j = 0; j < COMPSIZELIMIT; j++) stream << c.compSizeHistogram[j] << ",";
  }
  stream << std::endl;
}

// ---------------------------------------------------------------------------
// VPC (reference VPC.h:241-283, VPC.cpp:22-25, 72-330)
// ---------------------------------------------------------------------------
static void fail(const char *what, int rc, mpc_handle *h)
{
  const char *msg = mpc_last_error(h);
  printf("%s (%d): %s\n", what, rc, msg ? msg : "");
  exit(1);
}

VPC::VPC(std::string configPath) : m_Handle(nullptr)
{
  int rc = mpc_create_vpc(configPath.c_str(), -1, &m_Handle);
  if (rc == MPC_E_NOENT) {
    printf("Invalid File! \"%s\" is not valid path.\n", configPath.c_str());
    exit(1);
  }
  if (rc == MPC_E_PARSE) {
    printf("%s\n", mpc_last_error(nullptr));
    printf("Parsing ERROR! \"%s\" is not valid json file.\n", configPath.c_str());
    exit(1);
  }
  if (rc != MPC_OK) fail("VPC: cannot create the evaluator", rc, nullptr);
  mpc_info info;
  mpc_get_info(m_Handle, &info);
  m_LineSize = info.line_size;
  m_NumModules = info.num_modules;
  m_NumClusters = info.num_clusters;
  m_HistBins = info.hist_bins;
  m_Stat = new VPCResult((unsigned)m_LineSize, m_NumModules);
  m_Stat->CompressorName = "Contrastive Clustering Compressor";
  if (info.kernel_path == MPC_PATH_VPC_GENERIC)     // never silently: this path is some hundred times slower
    fprintf(stderr, "note: this configuration runs on the generic (slow, exact) kernel: %s\n", mpc_path_reason(m_Handle));
}

VPC::~VPC() { mpc_destroy(m_Handle); }

unsigned VPC::CompressLine(std::vector<uint8_t> &dataLine)
{
  if ((int)dataLine.size() != m_LineSize) {
    printf("VPC: line of %zu bytes, configuration lineSize is %d.\n", dataLine.size(), m_LineSize);
    exit(1);
  }
  if (LineBuffering()) {
    BufferLine(dataLine);
    return 0;
  }
  uint16_t bits = 0;
  int rc = mpc_compress_batch(m_Handle, dataLine.data(), 1, &bits, nullptr);
  if (rc != MPC_OK) fail("VPC::CompressLine", rc, m_Handle);
  return bits;
}

void VPC::CompressBatch(const uint8_t *lines, unsigned long long n)
{
  FlushLines();
  int rc = mpc_compress_batch(m_Handle, lines, n, nullptr, nullptr);
  if (rc != MPC_OK) fail("VPC::CompressBatch", rc, m_Handle);
}

unsigned long long VPC::CompressFile(const std::string &tracePath)
{
  FlushLines();
  uint64_t done = 0;
  const bool isLog = tracePath.size() > 4 && tracePath.compare(tracePath.size() - 4, 4, ".log") == 0;
  int rc = isLog ? mpc_compress_gpgpusim_log(m_Handle, tracePath.c_str(), nullptr, &done)
                 : mpc_compress_npy(m_Handle, tracePath.c_str(), 0, ~0ull, 1, &done);
  if (rc != MPC_OK) fail("VPC::CompressFile", rc, m_Handle);
  return done;
}

CompResult *VPC::GetResult()
{
  FlushLines();
  uint64_t len = 0;
  mpc_stats_len(m_Handle, &len);
  std::vector<uint64_t> v(len);
  int rc = mpc_stats_get(m_Handle, v.data(), v.size());
  if (rc != MPC_OK) fail("VPC::GetResult", rc, m_Handle);
  static_cast<VPCResult *>(m_Stat)->LoadVector(v.data(), m_NumModules, m_HistBins);
  return m_Stat;
}

}  // namespace comp
