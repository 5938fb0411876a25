#include "LoaderNPY.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace trace
{

LoaderNPY::LoaderNPY(const char *filePath) : Loader(filePath), m_CurrentLine(0), m_DataOffset(0), m_Loaded(false) { Reset(); }
LoaderNPY::LoaderNPY(const std::string filePath) : Loader(filePath), m_CurrentLine(0), m_DataOffset(0), m_Loaded(false) { Reset(); }

static void die(const std::string &path, const char *why)
{
  printf("Invalid File! \"%s\": %s\n", path.c_str(), why);
  exit(1);
}

void LoaderNPY::readHeader()
{
  if (!m_FileStream.is_open()) die(m_FilePath, "cannot be opened");
  m_FileStream.clear();
  m_FileStream.seekg(0);
  unsigned char pre[12];
  m_FileStream.read(reinterpret_cast<char *>(pre), 10);
  if (m_FileStream.gcount() != 10 || std::memcmp(pre, "\x93NUMPY", 6) != 0) die(m_FilePath, "not a .npy file");
  size_t hlen, off;
  if (pre[6] == 1) {
    hlen = (size_t)pre[8] | ((size_t)pre[9] << 8);
    off = 10;
  } else {
    m_FileStream.read(reinterpret_cast<char *>(pre + 10), 2);
    hlen = (size_t)pre[8] | ((size_t)pre[9] << 8) | ((size_t)pre[10] << 16) | ((size_t)pre[11] << 24);
    off = 12;
  }
  if (hlen > (1u << 20)) die(m_FilePath, "unreasonable .npy header");
  std::string hdr(hlen, '\0');
  m_FileStream.read(&hdr[0], (std::streamsize)hlen);
  if ((size_t)m_FileStream.gcount() != hlen) die(m_FilePath, "truncated .npy header");
  auto value_of = [&](const char *key) -> size_t {
    size_t p = hdr.find(key);
    if (p == std::string::npos) die(m_FilePath, "malformed .npy header");
    return hdr.find(':', p) + 1;
  };
  size_t p = value_of("'descr'");
  size_t q1 = hdr.find('\'', p), q2 = hdr.find('\'', q1 + 1);
  if (q1 == std::string::npos || q2 == std::string::npos) die(m_FilePath, "malformed .npy descr");
  std::string descr = hdr.substr(q1 + 1, q2 - q1 - 1);
  if (!(descr == "|u1" || descr == "<u1" || descr == "u1" || descr == "=u1")) die(m_FilePath, "dtype is not uint8");
  p = value_of("'fortran_order'");
  if (hdr.compare(hdr.find_first_not_of(' ', p), 5, "False") != 0) die(m_FilePath, "array is not C-order");
  p = value_of("'shape'");
  size_t a = hdr.find('(', p), b = hdr.find(')', a);
  if (a == std::string::npos || b == std::string::npos) die(m_FilePath, "malformed .npy shape");
  m_DataShape.clear();
  for (size_t i = a + 1; i < b;) {
    while (i < b && (hdr[i] < '0' || hdr[i] > '9')) i++;
    if (i >= b) break;
    uint64_t v = 0;
    while (i < b && hdr[i] >= '0' && hdr[i] <= '9') v = v * 10 + (uint64_t)(hdr[i++] - '0');
    m_DataShape.push_back(v);
  }
  if (m_DataShape.size() != 2) die(m_FilePath, "array is not 2-D [lines, line_size]");
  m_DataOffset = off + hlen;
}

void LoaderNPY::Reset()
{
  readHeader();
  m_CurrentLine = 0;
  m_Loaded = false;
  m_DataLines.clear();
}

MemReq_t *LoaderNPY::GetCacheline(MemReq_t *memReq)
{
  const uint64_t numTotalLines = m_DataShape[0], lineSize = m_DataShape[1];
  if (!m_Loaded) {
    m_DataLines.resize(numTotalLines * lineSize);
    m_FileStream.clear();
    m_FileStream.seekg((std::streamoff)m_DataOffset);
    m_FileStream.read(reinterpret_cast<char *>(m_DataLines.data()), (std::streamsize)m_DataLines.size());
    if ((uint64_t)m_FileStream.gcount() != m_DataLines.size()) die(m_FilePath, "file is shorter than its header says");
    m_Loaded = true;
  }
  memReq->addr = 0;
  memReq->rw = NA;
  memReq->reqSize = (uint32_t)lineSize;
  if (m_CurrentLine < numTotalLines)
    memReq->data.assign(m_DataLines.begin() + (std::ptrdiff_t)(m_CurrentLine * lineSize),
                        m_DataLines.begin() + (std::ptrdiff_t)((m_CurrentLine + 1) * lineSize));
  else
    memReq->data.assign(lineSize, 0);   // the reference reads out of bounds here (N == 0)
  m_CurrentLine++;
  memReq->isEnd = (m_CurrentLine >= numTotalLines);
  return memReq;
}

unsigned LoaderNPY::GetCachelineSize() { return (unsigned)m_DataShape[1]; }
unsigned long long LoaderNPY::GetNumLines() { return m_DataShape[0]; }

// Rows the per-line protocol still delivers: everything before the last row.
unsigned long long LoaderNPY::GetBatch(uint8_t *dst, unsigned long long maxLines)
{
  const uint64_t numTotalLines = m_DataShape[0], lineSize = m_DataShape[1];
  const uint64_t usable = numTotalLines > 0 ? numTotalLines - 1 : 0;
  if (m_CurrentLine >= usable) return 0;
  uint64_t take = usable - m_CurrentLine;
  if (take > maxLines) take = maxLines;
  m_FileStream.clear();
  m_FileStream.seekg((std::streamoff)(m_DataOffset + m_CurrentLine * lineSize));
  m_FileStream.read(reinterpret_cast<char *>(dst), (std::streamsize)(take * lineSize));
  if ((uint64_t)m_FileStream.gcount() != take * lineSize) die(m_FilePath, "file is shorter than its header says");
  m_CurrentLine += take;
  return take;
}

std::string LoaderNPY::GetStreamablePath() { return m_CurrentLine == 0 ? m_FilePath : std::string(); }

}  // namespace trace
