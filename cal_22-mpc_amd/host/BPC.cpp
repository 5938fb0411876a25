#include "BPC.h"

#include <cstdio>
#include <cstdlib>

#include "mpc_hip.h"

namespace comp
{

// reference BPC.h:35-84
void BPCResult::Print(std::string workloadName, std::string filePath)
{
  std::ofstream file;
  if (filePath != "")
    openForAppend(file, filePath,
                  "Workload,Original Size,Compressed Size,Compression Ratio,Total Words,Pattern0,Pattern1,Pattern2,Pattern3,"
                  "Pattern4,Pattern5,Pattern6,\n");
  std::ostream &stream = (filePath == "") ? std::cout : file;
  stream << workloadName << "," << OriginalSize << "," << CompressedSize << "," << mpctext::num(CompRatio) << ",";
  stream << TotalWords << ",";
  for (int i = 0; i < NUM_BPC_PATTERN; i++) stream << Counts[(size_t)i] << ",";
  stream << std::endl;
}

void BPCResult::LoadVector(const uint64_t *v)
{
  OriginalSize = v[1];
  CompressedSize = v[2];
  CompRatio = v[0] ? (double)OriginalSize / (double)CompressedSize : 0.0;
  TotalWords = v[3];
  for (int i = 0; i < NUM_BPC_PATTERN; i++) Counts[(size_t)i] = v[4 + i];
}

static void fail(const char *what, int rc, mpc_handle *h)
{
  const char *msg = mpc_last_error(h);
  printf("%s (%d): %s\n", what, rc, msg ? msg : "");
  exit(1);
}

BPC::BPC(unsigned lineSize) : m_Handle(nullptr), m_LineSize(lineSize)
{
  int rc = mpc_create_bpc(lineSize, -1, &m_Handle);
  if (rc != MPC_OK) fail("BPC: cannot create the evaluator", rc, nullptr);
  m_Stat = new BPCResult(lineSize);
  m_Stat->CompressorName = "Bit-Plane Compression";
}

BPC::~BPC() { mpc_destroy(m_Handle); }

unsigned BPC::CompressLine(std::vector<uint8_t> &dataLine)
{
  if (dataLine.size() != m_LineSize) {
    printf("BPC: line of %zu bytes, expected %u.\n", dataLine.size(), m_LineSize);
    exit(1);
  }
  if (LineBuffering()) {
    BufferLine(dataLine);
    return 0;
  }
  uint16_t bits = 0;
  int rc = mpc_compress_batch(m_Handle, dataLine.data(), 1, &bits, nullptr);
  if (rc != MPC_OK) fail("BPC::CompressLine", rc, m_Handle);
  return bits;
}

void BPC::CompressBatch(const uint8_t *lines, unsigned long long n)
{
  FlushLines();
  int rc = mpc_compress_batch(m_Handle, lines, n, nullptr, nullptr);
  if (rc != MPC_OK) fail("BPC::CompressBatch", rc, m_Handle);
}

unsigned long long BPC::CompressFile(const std::string &tracePath)
{
  FlushLines();
  uint64_t done = 0;
  const bool isLog = tracePath.size() > 4 && tracePath.compare(tracePath.size() - 4, 4, ".log") == 0;
  int rc = isLog ? mpc_compress_gpgpusim_log(m_Handle, tracePath.c_str(), nullptr, &done)
                 : mpc_compress_npy(m_Handle, tracePath.c_str(), 0, ~0ull, 1, &done);
  if (rc != MPC_OK) fail("BPC::CompressFile", rc, m_Handle);
  return done;
}

CompResult *BPC::GetResult()
{
  FlushLines();
  uint64_t v[11];
  int rc = mpc_stats_get(m_Handle, v, 11);
  if (rc != MPC_OK) fail("BPC::GetResult", rc, m_Handle);
  static_cast<BPCResult *>(m_Stat)->LoadVector(v);
  return m_Stat;
}

}  // namespace comp
