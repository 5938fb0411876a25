#include "LoaderGPGPU.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace trace
{
namespace gpgpusim
{

LoaderGPGPU::LoaderGPGPU(const char *filePath) : Loader(filePath), m_NumKeys(0), m_LineSize(0), m_Untouched(true) { Reset(); }
LoaderGPGPU::LoaderGPGPU(const std::string filePath) : Loader(filePath), m_NumKeys(0), m_LineSize(0), m_Untouched(true) { Reset(); }

unsigned LoaderGPGPU::GetCachelineSize()
{
  MemReqGPU_t first;
  GetCacheline(&first);
  const unsigned lineSize = first.reqSize;
  Reset();
  return lineSize;
}

// one record, field by field and little-endian as the reference reads it on x86
bool LoaderGPGPU::readRecord(MemReqGPU_t &r)
{
  unsigned char h[62];
  m_FileStream.read(reinterpret_cast<char *>(h), sizeof(h));
  if (m_FileStream.gcount() != (std::streamsize)sizeof(h)) return false;
  auto u32 = [&](int off) { uint32_t v; std::memcpy(&v, h + off, 4); return v; };
  auto u64 = [&](int off) { uint64_t v; std::memcpy(&v, h + off, 8); return v; };
  r.kernelID = h[0];
  r.mfType = (fetchTypeGPU)h[1];
  r.cycle = u64(2);
  r.tpc = u32(10);
  r.sid = u32(14);
  r.wid = u32(18);
  r.pc = u32(22);
  r.instCnt = u32(26);
  r.addr = u64(30);
  r.reqType = (reqTypeGPU)u32(38);
  r.row = u32(42);
  r.chip = u32(46);
  r.bank = u32(50);
  r.col = u32(54);
  r.reqSize = u32(58);
  if (r.reqSize > (1u << 20)) {
    printf("The GPGPU-sim trace file holds a %u-byte request: not a memory-line trace.\n", r.reqSize);
    exit(1);
  }
  r.data.resize(r.reqSize / sizeof(WORD_SIZE));
  m_FileStream.read(reinterpret_cast<char *>(r.data.data()), r.reqSize);
  return (uint32_t)m_FileStream.gcount() == r.reqSize;
}

MemReq_t *LoaderGPGPU::GetCacheline(MemReq_t *memReq)
{
  MemReqGPU_t *memReqGPU = static_cast<MemReqGPU_t *>(memReq);
  m_Untouched = false;
  memReqGPU->isEnd = !readRecord(*memReqGPU);
  return memReq;
}

unsigned long long LoaderGPGPU::GetNumLines()
{
  MemReqGPU_t req;
  unsigned long long numLines = 0;
  while (readRecord(req)) numLines++;
  Reset();
  return numLines;
}

unsigned long long LoaderGPGPU::GetBatch(uint8_t *dst, unsigned long long maxLines)
{
  if (m_LineSize == 0) {
    // line size = the first record's req_size (GetCachelineSize() rewinds, so ask before reading on)
    const std::streampos here = m_FileStream.tellg();
    m_LineSize = GetCachelineSize();
    m_FileStream.clear();
    m_FileStream.seekg(here);
  }
  MemReqGPU_t req;
  unsigned long long n = 0;
  m_Untouched = false;
  while (n < maxLines) {
    if (!readRecord(req)) break;
    if (!isEvaluated((uint32_t)req.reqType)) continue;
    if (req.reqSize != m_LineSize) {
      // the reference would hand the evaluator a line of another size (undefined behaviour for
      // VPC, VPC.cpp:332-364 reads config-L bytes); refuse instead
      printf("The GPGPU-sim trace mixes request sizes (%u after %u bytes): not supported.\n", req.reqSize, m_LineSize);
      exit(1);
    }
    std::memcpy(dst + n * m_LineSize, req.data.data(), m_LineSize);
    n++;
  }
  return n;
}

void LoaderGPGPU::Reset()
{
  m_FileStream.clear();
  m_FileStream.seekg(0);
  isFileValid();
  m_Untouched = true;
}

void LoaderGPGPU::isFileValid()
{
  if (!m_FileStream.is_open()) {
    printf("Failed to open a file. Check the path of the file.\n");
    exit(1);
  }
  m_NumKeys = 0;
  m_FileStream.read(reinterpret_cast<char *>(&m_NumKeys), 1);
  if (m_NumKeys != NUM_KEYS) {
    printf("The header of the GPGPU-sim trace file is not valid.\n");
    exit(1);
  }
  m_KeySizeList.clear();
  for (int i = 0; i < m_NumKeys; i++) {
    char buff[7] = {'\0'};
    unsigned char size = 0;
    m_FileStream.read(buff, 6);
    m_FileStream.read(reinterpret_cast<char *>(&size), 1);
    m_KeySizeList.insert(std::make_pair(std::string(buff), (int)size));
  }
  if (m_FileStream.eof() || m_FileStream.fail()) {
    printf("The header of the GPGPU-sim trace file is not valid.\n");
    exit(1);
  }
}

}  // namespace gpgpusim
}  // namespace trace
