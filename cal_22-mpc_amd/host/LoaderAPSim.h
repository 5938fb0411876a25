// LoaderAPSim.h -- APSim ".txt" traces of AXI-style four-channel data beats (reference
// src/loader/LoaderGPGPU.{h,cpp}, namespace trace::apsim, class LoaderGPGPU).
//
// File layout as the reference reads it (LoaderGPGPU.cpp:177-228, 469-510):
//   line 1   column names, comma separated; a column containing "last" makes it a READ trace,
//            one containing "strb" a WRITE trace (the first of the two decides), neither is an error
//   lines    cycle , clock , valid0..3 , data0..3 (64 hex digits = 32 bytes each, byte j = digits
//            2j, 2j+1) , ready0..3 , last0..3 (read) | strb0..3 (write, hex)          = 18 fields
// Behaviour kept:
//   * a row counts when clock != 0 and some channel has valid == 1 and ready == 1 (the handshake);
//     each such channel, in channel order, delivers its 32 data bytes (LoaderGPGPU.cpp:230-288, 451-458);
//   * line size 32 (what the driver asks for: main.cpp:23, 81): one line per handshaking channel;
//     line size 64: two consecutive beats of the SAME channel form a line, first beat first
//     (:330-441); beats still waiting for their partner at the end of the file are dropped;
//   * std::getline followed by an eof() test (:179-182): a final row without a newline is dropped;
//   * the 32-byte form reports rw = READ and reqSize = 64 for every line whatever the trace is
//     (:259, :264) -- kept, nothing on the evaluation path reads them.
// Deviation: a row with fewer than 18 fields or malformed numbers ends the program with a message
// (the reference indexes past the end of its vector / lets std::stoi throw).
#ifndef MPC_HOST_LOADERAPSIM_H
#define MPC_HOST_LOADERAPSIM_H

#include <cstring>
#include <queue>

#include "Loader.h"

namespace trace
{
namespace apsim
{

#define NUM_CH 4
#define BURST_LEN 2

struct DatasetAttr {
  uint64_t cycle;
  uint8_t clock;
  uint8_t valid[NUM_CH];
  uint8_t ready[NUM_CH];
  uint8_t last[NUM_CH];
  uint32_t strb[NUM_CH];
  uint8_t data[NUM_CH][ACCESS_GRAN];

  void Reset()
  {
    cycle = 0;
    clock = 0;
    memset(valid, 0, sizeof(valid));
    memset(ready, 0, sizeof(ready));
    memset(last, 0, sizeof(last));
    memset(strb, 0, sizeof(strb));
    memset(data, 0, sizeof(data));
  }
};

struct MemReqGPU_t : public MemReq_t {
  uint64_t cycle;
  uint8_t ch;

  MemReqGPU_t() { Reset(); }
  virtual void Reset()
  {
    MemReq_t::Reset();
    cycle = 0;
    ch = 0;
  }
  void Set(MemReqGPU_t &other) { *this = other; }
};

class LoaderGPGPU : public Loader
{
public:
  LoaderGPGPU(const char *filePath);
  LoaderGPGPU(const std::string filePath);
  LoaderGPGPU(const char *filePath, const unsigned lineSize);
  LoaderGPGPU(const std::string filePath, const unsigned lineSize);

  virtual MemReq_t *GetCacheline(MemReq_t *memReq);    // memReq must be an apsim::MemReqGPU_t
  virtual unsigned GetCachelineSize();
  bool ReadLine(DatasetAttr &datasetAttr);              // one row of the file; false at its end
  virtual unsigned long long GetNumLines();             // lines GetCacheline delivers; rewinds
  virtual void Reset();

  // ADDITIVE: the lines GetCacheline() would still deliver, packed
  virtual unsigned long long GetBatch(uint8_t *dst, unsigned long long maxLines);
  virtual bool SupportsBatch() { return true; }

  rw_t GetRW() const { return m_RW; }

private:
  bool fill();                                           // reads rows until a line is queued (or the end marker)
  std::vector<uint8_t> getHandshakingChannels(DatasetAttr &datasetAttr);
  void isFileValid();

protected:
  rw_t m_RW;
  std::queue<MemReqGPU_t> m_MemReqQueue;
  std::queue<MemReqGPU_t> m_MemReqChQueue[NUM_CH];
  const unsigned m_LineSize;
};

}  // namespace apsim
}  // namespace trace

#endif
