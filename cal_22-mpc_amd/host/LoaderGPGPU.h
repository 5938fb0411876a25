// LoaderGPGPU.h -- GPGPU-Sim memory-request traces (".log"; reference
// src/loader/LoaderGPGPU.{h,cpp}, namespace trace::gpgpusim).
//
// File layout as the reference reads it (LoaderGPGPU.cpp:26-47, 93-119):
//   1 byte   number of keys, must be NUM_KEYS (17)
//   17 x     { 6 bytes key name, 1 byte field size }            (read, not interpreted)
//   records  kid u8 | mf_type u8 | cycle u64 | tpc u32 | sid u32 | wid u32 | pc u32 |
//            inst_cnt u32 | mem_addr u64 | req_type u32 | row u32 | chip u32 | bank u32 |
//            col u32 | req_size u32 | req_size bytes of data          (62 + req_size bytes)
// Behaviour kept: GetCachelineSize() is the req_size of the FIRST record; a record is
// delivered when all of it could be read, isEnd is raised by the read that runs into the
// end of the file (so an incomplete trailing record is dropped and every complete one is
// delivered); the driver, not the loader, keeps only GLOBAL_ACC_R / GLOBAL_ACC_W requests
// (main.cpp:222-224).  The additive GetBatch() applies that same filter and packs the
// kept lines, which is what the batch driver needs.
#ifndef MPC_HOST_LOADERGPGPU_H
#define MPC_HOST_LOADERGPGPU_H

#include <map>

#include "Loader.h"

namespace trace
{
namespace gpgpusim
{

#define NUM_KEYS 17

// GPGPU-sim mem request type (reference LoaderGPGPU.h:17-28)
enum reqTypeGPU {
  GLOBAL_ACC_R = 0,
  LOCAL_ACC_R = 1,
  CONST_ACC_R = 2,
  TEXTURE_ACC_R = 3,
  GLOBAL_ACC_W = 4,
  LOCAL_ACC_W = 5,
  L1_WRBK_ACC = 6,
  L2_WRBK_ACC = 7,
  INST_ACC_R = 8,
};

// GPGPU-sim mem fetch type (reference LoaderGPGPU.h:31-37)
enum fetchTypeGPU {
  READ_REQUEST = 0,
  WRITE_REQUEST = 1,
  READ_REPLY = 2,
  WRITE_ACK = 3,
};

struct MemReqGPU_t : public MemReq_t {
  uint8_t kernelID;
  uint64_t cycle;
  uint32_t tpc, sid, wid, pc, instCnt;
  reqTypeGPU reqType;
  fetchTypeGPU mfType;
  uint32_t row, chip, bank, col;   // DRAM command

  MemReqGPU_t() { Reset(); }
  virtual void Reset()
  {
    MemReq_t::Reset();
    kernelID = 0;
    cycle = 0;
    tpc = sid = wid = pc = instCnt = 0;
    reqType = GLOBAL_ACC_R;
    mfType = READ_REQUEST;
    row = chip = bank = col = 0;
  }
};

class LoaderGPGPU : public Loader
{
public:
  LoaderGPGPU(const char *filePath);
  LoaderGPGPU(const std::string filePath);

  virtual unsigned GetCachelineSize();                 // req_size of the first record; rewinds
  virtual MemReq_t *GetCacheline(MemReq_t *memReq);    // memReq must be a MemReqGPU_t
  virtual unsigned long long GetNumLines();            // complete records of any request type; rewinds
  virtual void Reset();

  // ADDITIVE: the GLOBAL_ACC_R / GLOBAL_ACC_W lines of the records not yet delivered, packed
  virtual unsigned long long GetBatch(uint8_t *dst, unsigned long long maxLines);
  virtual bool SupportsBatch() { return true; }
  // the library streams .log files itself (mpc_compress_gpgpusim_log) while nothing has been read yet
  virtual std::string GetStreamablePath() { return m_Untouched ? m_FilePath : std::string(); }

  static bool isEvaluated(uint32_t reqType) { return reqType == GLOBAL_ACC_R || reqType == GLOBAL_ACC_W; }

private:
  void isFileValid();
  bool readRecord(MemReqGPU_t &req);   // false: ran into the end of the file

  uint8_t m_NumKeys;
  std::map<std::string, int> m_KeySizeList;
  unsigned m_LineSize;                 // cached by GetBatch
  bool m_Untouched;                    // no record delivered since the last Reset()
};

}  // namespace gpgpusim
}  // namespace trace

#endif
