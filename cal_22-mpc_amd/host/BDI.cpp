#include "BDI.h"

#include <cstdio>
#include <cstdlib>

#include "mpc_hip.h"

namespace comp
{

void BDIResult::Print(std::string workloadName, std::string filePath)
{
  std::ofstream file;
  if (filePath != "")
    openForAppend(file, filePath,
                  "Workload,Original Size,Compressed Size,Compression Ratio,Zeros,Repeated,B8D1,B8D2,B8D4,B4D1,B4D2,"
                  "B2D1,Uncompressed,\n");
  std::ostream &stream = (filePath == "") ? std::cout : file;
  stream << workloadName << "," << OriginalSize << "," << CompressedSize << "," << mpctext::num(CompRatio) << ",";
  for (int i = 0; i < 9; i++) stream << Counts[(size_t)i] << ",";
  stream << std::endl;
}

void BDIResult::LoadVector(const uint64_t *v)
{
  OriginalSize = v[1];
  CompressedSize = v[2];
  CompRatio = v[0] ? (double)OriginalSize / (double)CompressedSize : 0.0;
  for (int i = 0; i < 9; i++) Counts[(size_t)i] = v[3 + i];
}

static void fail(const char *what, int rc, mpc_handle *h)
{
  const char *msg = mpc_last_error(h);
  printf("%s (%d): %s\n", what, rc, msg ? msg : "");
  exit(1);
}

BDI::BDI(unsigned lineSize) : m_Handle(nullptr), m_LineSize(lineSize)
{
  int rc = mpc_create_bdi(lineSize, -1, &m_Handle);
  if (rc != MPC_OK) fail("BDI: cannot create the evaluator", rc, nullptr);
  m_Stat = new BDIResult(lineSize);
  m_Stat->CompressorName = "Base-Delta Immediate";
}

BDI::~BDI() { mpc_destroy(m_Handle); }

unsigned BDI::CompressLine(std::vector<uint8_t> &dataLine)
{
  if (dataLine.size() != m_LineSize) {
    printf("BDI: line of %zu bytes, expected %u.\n", dataLine.size(), m_LineSize);
    exit(1);
  }
  if (LineBuffering()) {
    BufferLine(dataLine);
    return 0;
  }
  uint16_t bits = 0;
  int rc = mpc_compress_batch(m_Handle, dataLine.data(), 1, &bits, nullptr);
  if (rc != MPC_OK) fail("BDI::CompressLine", rc, m_Handle);
  return bits;
}

void BDI::CompressBatch(const uint8_t *lines, unsigned long long n)
{
  FlushLines();
  int rc = mpc_compress_batch(m_Handle, lines, n, nullptr, nullptr);
  if (rc != MPC_OK) fail("BDI::CompressBatch", rc, m_Handle);
}

unsigned long long BDI::CompressFile(const std::string &tracePath)
{
  FlushLines();
  uint64_t done = 0;
  const bool isLog = tracePath.size() > 4 && tracePath.compare(tracePath.size() - 4, 4, ".log") == 0;
  int rc = isLog ? mpc_compress_gpgpusim_log(m_Handle, tracePath.c_str(), nullptr, &done)
                 : mpc_compress_npy(m_Handle, tracePath.c_str(), 0, ~0ull, 1, &done);
  if (rc != MPC_OK) fail("BDI::CompressFile", rc, m_Handle);
  return done;
}

CompResult *BDI::GetResult()
{
  FlushLines();
  uint64_t v[12];
  int rc = mpc_stats_get(m_Handle, v, 12);
  if (rc != MPC_OK) fail("BDI::GetResult", rc, m_Handle);
  static_cast<BDIResult *>(m_Stat)->LoadVector(v);
  return m_Stat;
}

}  // namespace comp
