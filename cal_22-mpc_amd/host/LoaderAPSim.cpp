#include "LoaderAPSim.h"

#include <cstdio>
#include <cstdlib>

#include "utils.h"

namespace trace
{
namespace apsim
{

static void checkLineSize(unsigned lineSize)
{
  if (!(lineSize == ACCESS_GRAN || lineSize == ACCESS_GRAN * BURST_LEN)) {    // the reference asserts (LoaderGPGPU.cpp:153-154)
    printf("Invalid cache-line size\n");
    exit(1);
  }
}

LoaderGPGPU::LoaderGPGPU(const char *filePath) : Loader(filePath), m_RW(NA), m_LineSize(ACCESS_GRAN) { Reset(); }
LoaderGPGPU::LoaderGPGPU(const std::string filePath) : Loader(filePath), m_RW(NA), m_LineSize(ACCESS_GRAN) { Reset(); }
LoaderGPGPU::LoaderGPGPU(const char *filePath, const unsigned lineSize) : Loader(filePath), m_RW(NA), m_LineSize(lineSize) { Reset(); }
LoaderGPGPU::LoaderGPGPU(const std::string filePath, const unsigned lineSize) : Loader(filePath), m_RW(NA), m_LineSize(lineSize) { Reset(); }

unsigned LoaderGPGPU::GetCachelineSize() { return m_LineSize; }

unsigned long long LoaderGPGPU::GetNumLines()
{
  MemReqGPU_t req;
  unsigned long long numLines = 0;
  while (1) {
    GetCacheline(&req);
    if (req.isEnd) break;
    numLines++;
  }
  Reset();
  return numLines;
}

void LoaderGPGPU::Reset()
{
  checkLineSize(m_LineSize);
  m_FileStream.clear();
  m_FileStream.seekg(0);
  while (!m_MemReqQueue.empty()) m_MemReqQueue.pop();
  for (int ch = 0; ch < NUM_CH; ch++)
    while (!m_MemReqChQueue[ch].empty()) m_MemReqChQueue[ch].pop();
  isFileValid();
}

static bool parseUnsigned(const std::string &s, int base, unsigned long long &out)
{
  if (s.empty()) return false;
  char *end = nullptr;
  out = strtoull(s.c_str(), &end, base);
  return end != s.c_str();
}

static int hexDigit(char c)
{
  if (c >= '0' && c <= '9') return c - '0';
  if (c >= 'a' && c <= 'f') return c - 'a' + 10;
  if (c >= 'A' && c <= 'F') return c - 'A' + 10;
  return -1;
}

bool LoaderGPGPU::ReadLine(DatasetAttr &a)
{
  std::string line;
  std::getline(m_FileStream, line);
  if (m_FileStream.eof()) return false;       // also drops a last row that has no newline (LoaderGPGPU.cpp:181)
  std::vector<std::string> f = mpctext::split(line, ",");
  bool ok = f.size() >= 18;
  unsigned long long v = 0;
  if (ok) {
    ok = parseUnsigned(f[0], 10, v);
    a.cycle = v;
    ok = ok && parseUnsigned(f[1], 10, v);
    a.clock = (uint8_t)v;
  }
  for (int i = 0; ok && i < NUM_CH; i++) {
    ok = ok && parseUnsigned(f[2 + i], 10, v);
    a.valid[i] = (uint8_t)v;
    ok = ok && parseUnsigned(f[10 + i], 10, v);
    a.ready[i] = (uint8_t)v;
    if (m_RW == READ) {
      ok = ok && parseUnsigned(f[14 + i], 10, v);
      a.last[i] = (uint8_t)v;
    } else {
      ok = ok && parseUnsigned(f[14 + i], 16, v);
      a.strb[i] = (uint8_t)v;                 // the reference keeps the low byte only (LoaderGPGPU.cpp:217)
    }
    const std::string &hex = f[6 + i];
    ok = ok && hex.size() >= 2 * ACCESS_GRAN;
    for (int j = 0; ok && j < ACCESS_GRAN; j++) {
      const int hi = hexDigit(hex[2 * j]), lo = hexDigit(hex[2 * j + 1]);
      ok = hi >= 0 && lo >= 0;
      if (ok) a.data[i][j] = (uint8_t)((hi << 4) | lo);
    }
  }
  if (!ok) {
    printf("A row of the GPU traffic file is not valid.\n");
    exit(1);
  }
  return true;
}

std::vector<uint8_t> LoaderGPGPU::getHandshakingChannels(DatasetAttr &a)
{
  std::vector<uint8_t> channels;
  for (int i = 0; i < NUM_CH; i++)
    if (a.valid[i] == 1 && a.ready[i] == 1) channels.push_back((uint8_t)i);
  return channels;
}

// Reads rows until at least one line is queued; at the end of the file queues the end marker.
bool LoaderGPGPU::fill()
{
  DatasetAttr a;
  a.Reset();
  while (true) {
    if (!ReadLine(a)) {
      MemReqGPU_t end;
      end.Reset();
      end.isEnd = true;
      m_MemReqQueue.push(end);
      return false;
    }
    if (!a.clock) continue;
    if (!a.valid[0] && !a.valid[1] && !a.valid[2] && !a.valid[3]) continue;
    std::vector<uint8_t> channels = getHandshakingChannels(a);
    if (channels.empty()) continue;
    bool queued = false;
    for (uint8_t ch : channels) {
      MemReqGPU_t beat;
      beat.rw = READ;
      beat.cycle = a.cycle;
      beat.ch = ch;
      beat.reqSize = ACCESS_GRAN * BURST_LEN;
      beat.data.assign(a.data[ch], a.data[ch] + ACCESS_GRAN);
      beat.isEnd = false;
      if (m_LineSize == ACCESS_GRAN) {
        m_MemReqQueue.push(beat);
        queued = true;
      } else {
        m_MemReqChQueue[ch].push(beat);
      }
    }
    if (m_LineSize != ACCESS_GRAN) {
      // a channel with two beats waiting: one 64-byte line, first beat first (LoaderGPGPU.cpp:372-410)
      for (int ch = 0; ch < NUM_CH; ch++) {
        if (m_MemReqChQueue[ch].size() == BURST_LEN) {
          MemReqGPU_t line;
          line.ch = (uint8_t)ch;
          line.rw = m_RW;
          line.reqSize = ACCESS_GRAN * BURST_LEN;
          line.isEnd = false;
          const MemReqGPU_t first = m_MemReqChQueue[ch].front();
          m_MemReqChQueue[ch].pop();
          const MemReqGPU_t second = m_MemReqChQueue[ch].front();
          m_MemReqChQueue[ch].pop();
          line.data = first.data;
          line.data.insert(line.data.end(), second.data.begin(), second.data.end());
          line.cycle = second.cycle;
          m_MemReqQueue.push(line);
          queued = true;
        }
      }
    }
    if (queued) return true;
  }
}

MemReq_t *LoaderGPGPU::GetCacheline(MemReq_t *memReq)
{
  if (m_MemReqQueue.empty()) fill();
  MemReqGPU_t *out = static_cast<MemReqGPU_t *>(memReq);
  *out = m_MemReqQueue.front();
  m_MemReqQueue.pop();
  return memReq;
}

unsigned long long LoaderGPGPU::GetBatch(uint8_t *dst, unsigned long long maxLines)
{
  unsigned long long n = 0;
  while (n < maxLines) {
    if (m_MemReqQueue.empty()) fill();
    if (m_MemReqQueue.front().isEnd) break;     // the end marker stays queued: every later call returns 0
    const MemReqGPU_t &r = m_MemReqQueue.front();
    std::memcpy(dst + n * m_LineSize, r.data.data(), m_LineSize);
    m_MemReqQueue.pop();
    n++;
  }
  return n;
}

void LoaderGPGPU::isFileValid()
{
  if (!m_FileStream.is_open()) {
    printf("Failed to open a file. Check the path of the file.\n");
    exit(1);
  }
  std::string firstLine;
  std::getline(m_FileStream, firstLine);
  m_RW = NA;
  for (const std::string &col : mpctext::split(firstLine, ",")) {
    if (col.find("last") != std::string::npos) { m_RW = READ; break; }
    if (col.find("strb") != std::string::npos) { m_RW = WRITE; break; }
  }
  if (m_RW == NA || m_FileStream.eof()) {
    printf("The header of the GPU traffic file is not valid.\n");
    exit(1);
  }
}

}  // namespace apsim
}  // namespace trace
