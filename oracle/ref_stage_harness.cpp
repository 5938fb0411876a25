// ref_stage_harness.cpp -- ORACLE-SIDE test infrastructure (NOT product code).
//
// A thin extern "C" driver around the reference's OWN, UNMODIFIED stage classes
// so the C restatement in mpc_oracle.c can be checked against them.  It is
// linked (oracle/Makefile, target _ref) with exactly these reference
// translation units, compiled where they lie under /root/reference:
//
//   src/compressor/VPCmodules/PredictorModule.cpp   
//   src/compressor/VPCmodules/ResidueModule.cpp     (8a-6, 8a-13)
//   src/compressor/VPCmodules/XORModule.cpp         (8a-8)
//   src/compressor/VPCmodules/ScanModule.cpp        (8a-9)
//   src/compressor/VPCmodules/FPCModule.cpp         (8a-11)
//
// Those five include only each other's headers and the C++ standard library.
// Every other file on the path (VPC.cpp, BDI.cpp, AllZeroModule.cpp,
// AllWordSameModule.cpp, BitplaneModule.cpp) reaches <strutil.h> /
// <json/json.h> through ../Compressor.h; those libraries are not in this image
// and no stand-ins are written for them, so those files are NOT built.  The
// one stage in the middle of the chain that cannot be built, the bit-plane
// transpose (BitplaneModule.cpp:7-51), is done here by make_bitplane() into
// the reference's own `Binary` container.
//
// The output library lives in oracle/_ref/ (git-ignored) and is only ever
// loaded by tests/.
#include <cstdint>
#include <vector>

#include "PredictorModule.h"
#include "ResidueModule.h"
#include "XORModule.h"
#include "ScanModule.h"
#include "FPCModule.h"

namespace {

comp::PredictorModule *make_predictor(int kind, int root, int L, const int *base,
                                      const float *weight, const int *diff)
{
  switch (kind) {
  case 0: {
    std::vector<int> b(base, base + L);
    std::vector<float> w(weight, weight + L);
    return new comp::WeightBasePredictor(root, L, b, w);
  }
  case 1: {
    std::vector<int> b(base, base + L);
    std::vector<int> d(diff, diff + L);
    return new comp::DiffBasePredictor(root, L, b, d);
  }
  case 2:
    return new comp::OneBasePredictor(root, L);
  default:
    return new comp::ConsecutiveBasePredictor(root, L);
  }
}

// bitplane[r][c] = bit (7-r) of residue c, stored one byte per bit.
comp::Binary make_bitplane(comp::Symbol &residue)
{
  const int L = residue.GetCachelineSize();
  comp::Binary bp;
  bp.SetSize(8, L);
  for (int c = 0; c < L; c++)
    for (int r = 0; r < 8; r++)
      bp[r][c] = (uint8_t)((residue[c] >> (7 - r)) & 1);
  bp.SetRootIndex(residue.GetRootIndex());
  return bp;
}

}  // namespace

extern "C" {

void ref_predict(int kind, int root, int L, const int *base, const float *weight,
                 const int *diff, const uint8_t *line, uint8_t *out)
{
  comp::PredictorModule *p = make_predictor(kind, root, L, base, weight, diff);
  std::vector<uint8_t> v(line, line + L);
  comp::Symbol s = p->PredictLine(v);
  for (int i = 0; i < L; i++) out[i] = s[i];
  delete p;
}

void ref_residue(int kind, int root, int L, const int *base, const float *weight,
                 const int *diff, const uint8_t *line, uint8_t *out)
{
  comp::PredictorModule *p = make_predictor(kind, root, L, base, weight, diff);
  comp::ResidueModule rm(p);
  std::vector<uint8_t> v(line, line + L);
  comp::Symbol s = rm.ProcessLine(v);
  for (int i = 0; i < L; i++) out[i] = s[i];
  delete p;
}

void ref_mae_mse(int kind, int root, int L, const int *base, const float *weight,
                 const int *diff, const uint8_t *line, double *mae, double *mse)
{
  comp::PredictorModule *p = make_predictor(kind, root, L, base, weight, diff);
  comp::ResidueModule rm(p);
  std::vector<uint8_t> v(line, line + L);
  *mae = rm.GetMAE(v);
  *mse = rm.GetMSE(v);
  delete p;
}

// residue -> bitplane -> XOR -> scan; out[r] bit (15-c) = scanned[r][c]
void ref_scanned(int kind, int root, int L, const int *base, const float *weight,
                 const int *diff, int consecutive_xor, int table_size, const int *rows,
                 const int *cols, const uint8_t *line, uint16_t *out)
{
  comp::PredictorModule *p = make_predictor(kind, root, L, base, weight, diff);
  comp::ResidueModule rm(p);
  comp::XORModule xm(consecutive_xor != 0);
  comp::ScanModule sm(table_size, std::vector<int>(rows, rows + table_size),
                      std::vector<int>(cols, cols + table_size));
  std::vector<uint8_t> v(line, line + L);
  comp::Symbol res = rm.ProcessLine(v);
  comp::Binary bp = make_bitplane(res);
  comp::Binary bx = xm.ProcessLine(bp);
  comp::Binary sc = sm.ProcessLine(bx);
  for (int r = 0; r < sc.GetRowSize(); r++) {
    uint16_t w = 0;
    for (int c = 0; c < 16; c++)
      if (sc[r][c]) w |= (uint16_t)(1u << (15 - c));
    out[r] = w;
  }
  delete p;
}

int ref_fpc_size(const uint16_t *scanned, int rows)
{
  comp::Binary b;
  b.SetSize(rows, 16);
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < 16; c++)
      b[r][c] = (uint8_t)((scanned[r] >> (15 - c)) & 1);
  comp::FPCModule enc;
  return enc.ProcessLine(b);
}

}  // extern "C"
