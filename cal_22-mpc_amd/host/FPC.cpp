#include "FPC.h"

#include <cstdio>
#include <cstdlib>

#include "mpc_hip.h"

namespace comp
{

// reference FPC.h:40-84
void FPCResult::Print(std::string workloadName, std::string filePath)
{
  std::ofstream file;
  if (filePath != "")
    openForAppend(file, filePath,
                  "Workload,Original Size,Compressed Size,Compression Ratio,Total Words,Prefix0,Prefix1,Prefix2,Prefix3,"
                  "Prefix4,Prefix5,Prefix6,Prefix7,\n");
  std::ostream &stream = (filePath == "") ? std::cout : file;
  stream << workloadName << "," << OriginalSize << "," << CompressedSize << "," << mpctext::num(CompRatio) << ",";
  stream << TotalWords << ",";
  for (int i = 0; i < NUM_FPC_PATTERN; i++) stream << Counts[(size_t)i] << ",";
  stream << std::endl;
}

void FPCResult::LoadVector(const uint64_t *v)
{
  OriginalSize = v[1];
  CompressedSize = v[2];
  CompRatio = v[0] ? (double)OriginalSize / (double)CompressedSize : 0.0;
  TotalWords = 0;
  for (int i = 0; i < NUM_FPC_PATTERN; i++) {
    Counts[(size_t)i] = v[3 + i];
    TotalWords += v[3 + i];
  }
}

static void fail(const char *what, int rc, mpc_handle *h)
{
  const char *msg = mpc_last_error(h);
  printf("%s (%d): %s\n", what, rc, msg ? msg : "");
  exit(1);
}

FPC::FPC(unsigned lineSize) : m_Handle(nullptr), m_LineSize(lineSize)
{
  int rc = mpc_create_fpc(lineSize, -1, &m_Handle);
  if (rc != MPC_OK) fail("FPC: cannot create the evaluator", rc, nullptr);
  m_Stat = new FPCResult(lineSize);
  m_Stat->CompressorName = "Frequent Pattern Compression";
}

FPC::~FPC() { mpc_destroy(m_Handle); }

unsigned FPC::CompressLine(std::vector<uint8_t> &dataLine)
{
  if (dataLine.size() != m_LineSize) {
    printf("FPC: line of %zu bytes, expected %u.\n", dataLine.size(), m_LineSize);
    exit(1);
  }
  if (LineBuffering()) {
    BufferLine(dataLine);
    return 0;
  }
  uint16_t bits = 0;
  int rc = mpc_compress_batch(m_Handle, dataLine.data(), 1, &bits, nullptr);
  if (rc != MPC_OK) fail("FPC::CompressLine", rc, m_Handle);
  return bits;
}

void FPC::CompressBatch(const uint8_t *lines, unsigned long long n)
{
  FlushLines();
  int rc = mpc_compress_batch(m_Handle, lines, n, nullptr, nullptr);
  if (rc != MPC_OK) fail("FPC::CompressBatch", rc, m_Handle);
}

unsigned long long FPC::CompressFile(const std::string &tracePath)
{
  FlushLines();
  uint64_t done = 0;
  const bool isLog = tracePath.size() > 4 && tracePath.compare(tracePath.size() - 4, 4, ".log") == 0;
  int rc = isLog ? mpc_compress_gpgpusim_log(m_Handle, tracePath.c_str(), nullptr, &done)
                 : mpc_compress_npy(m_Handle, tracePath.c_str(), 0, ~0ull, 1, &done);
  if (rc != MPC_OK) fail("FPC::CompressFile", rc, m_Handle);
  return done;
}

CompResult *FPC::GetResult()
{
  FlushLines();
  uint64_t v[11];
  int rc = mpc_stats_get(m_Handle, v, 11);
  if (rc != MPC_OK) fail("FPC::GetResult", rc, m_Handle);
  static_cast<FPCResult *>(m_Stat)->LoadVector(v);
  return m_Stat;
}

}  // namespace comp
