/*
 * mpc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See mpc_oracle.h for scope and pinning status.  Written from a reading of
 * the reference sources; deliberately simple (byte arrays, one bit per cell
 * like the reference's Binary container) so it is easy to audit against them.
 *
 * Citations are relative to /root/reference/src/compressor/.
 */
#include "mpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BYTE 8 /* CompResult.h:13 */

/* ------------------------------------------------------------------------ */
/* config validation: what the reference needs to run without UB            */
/* ------------------------------------------------------------------------ */

/* WeightBasePredictor ctor, PredictorModule.cpp:21-34: (int)log2f(weight) */
static int weight_shift(float w) { return (int)log2f(w); }

int mpc_o_vpc_validate(const mpc_o_config *cfg)
{
  const int M = cfg->num_modules, L = cfg->line_size;
  if (M < 1 || M > MPC_O_MAX_MODULES) return -1;
  if (L < 4 || L > MPC_O_MAX_LINE || (L % 2)) return -2;
  /* VPC.cpp:312,318: the dispatch pointer is only set by AllZero/AllWordSame;
   * VPC.cpp:336,353,376: module 0 is cast to AllZero, module 1 to AllWordSame,
   * the rest to PredComp.                                                   */
  if (cfg->modules[0].kind != MPC_O_ALLZERO) return -3;
  for (int i = 1; i < M; i++) {
    const mpc_o_module *m = &cfg->modules[i];
    if (m->kind == MPC_O_ALLZERO) return -4;
    if (m->kind == MPC_O_ALLWORDSAME && i != 1) return -5;
    if (m->kind != MPC_O_PREDCOMP) continue;
    if (m->root < 0 || m->root >= L) return -6;
    /* PredictorModule.cpp:167: inputLine[i-1] with i==0 unless root==0 */
    if (m->pred_kind == MPC_O_PRED_CONSEC && m->root != 0) return -7;
    if (m->pred_kind == MPC_O_PRED_WEIGHT || m->pred_kind == MPC_O_PRED_DIFF) {
      for (int j = 0; j < L; j++) {
        if (j == m->root) continue;
        if (m->base[j] < 0 || m->base[j] >= L) return -8;
        if (m->pred_kind == MPC_O_PRED_WEIGHT) {
          float w = m->weight[j];
          if (!(w > 0.0f) || isinf(w)) return -9;
          int s = weight_shift(w);
          if (s < -31 || s > 31) return -9;
        }
      }
    }
    if (m->table_size < 0 || m->table_size > 8 * L) return -10;
    for (int j = 0; j < m->table_size; j++) {
      if (m->rows[j] < 0 || m->rows[j] >= BYTE) return -11;
      if (m->cols[j] < 0 || m->cols[j] >= L) return -12;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* predictors: VPCmodules/PredictorModule.cpp                               */
/* ------------------------------------------------------------------------ */

void mpc_o_predict(const mpc_o_module *m, int L, const uint8_t *line, uint8_t *pred)
{
  switch (m->pred_kind) {
  case MPC_O_PRED_WEIGHT: /* PredictorModule.cpp:37-68 */
    for (int i = 0; i < L; i++) {
      if (i == m->root) {
        pred[i] = line[i];
      } else {
        uint8_t base = line[m->base[i]];
        int s = weight_shift(m->weight[i]);
        if (s < 0)
          pred[i] = (uint8_t)(base >> abs(s));
        else
          pred[i] = (uint8_t)(base << s);
      }
    }
    break;
  case MPC_O_PRED_DIFF: /* PredictorModule.cpp:82-110 */
    for (int i = 0; i < L; i++) {
      if (i == m->root)
        pred[i] = line[i];
      else
        pred[i] = (uint8_t)((uint8_t)m->diff[i] + line[m->base[i]]);
    }
    break;
  case MPC_O_PRED_ONEBASE: /* PredictorModule.cpp:113-130 */
    for (int i = 0; i < L; i++)
      pred[i] = line[m->root];
    break;
  case MPC_O_PRED_CONSEC: { /* PredictorModule.cpp:133-173, mb_Byteplane=true */
    uint8_t inp[MPC_O_MAX_LINE];
    int idx = 0;
    for (int plane = 3; plane >= 0; plane--)
      for (int i = plane; i < L; i += 4)
        inp[idx++] = line[i];
    for (int i = 0; i < L; i++) {
      if (i == m->root)
        pred[i] = inp[i];
      else
        pred[i] = inp[i - 1];
    }
    break;
  }
  }
}

/* ResidueModule::ProcessLine, ResidueModule.cpp:12-41: root first, then the
 * other positions in order, each (line - predicted) mod 256.                */
void mpc_o_residue(const mpc_o_module *m, int L, const uint8_t *line, uint8_t *res)
{
  uint8_t pred[MPC_O_MAX_LINE];
  mpc_o_predict(m, L, line, pred);
  res[0] = line[m->root];
  int j = 1;
  for (int i = 0; i < L; i++) {
    if (i == m->root) continue;
    res[j++] = (uint8_t)(line[i] - pred[i]);
  }
}

/* BitplaneModule.cpp:7-51, XORModule.cpp:5-23, ScanModule.cpp:6-22 */
void mpc_o_scanned(const mpc_o_module *m, int L, const uint8_t *line, uint16_t *scanned)
{
  uint8_t res[MPC_O_MAX_LINE];
  static _Thread_local uint8_t plane[BYTE][MPC_O_MAX_LINE];
  static _Thread_local uint8_t x[BYTE][MPC_O_MAX_LINE];

  mpc_o_residue(m, L, line, res);

  /* bitplane[r][c] = bit (7-r) of residue c: row 0 is the MSB plane */
  for (int c = 0; c < L; c++)
    for (int r = 0; r < BYTE; r++)
      plane[r][c] = (res[c] >> ((BYTE - 1) - r)) & 1;

  /* XOR: rows and columns both start at 1 (row 0, column 0 copied) */
  for (int r = 0; r < BYTE; r++)
    for (int c = 0; c < L; c++)
      x[r][c] = plane[r][c];
  for (int r = 1; r < BYTE; r++)
    for (int c = 1; c < L; c++)
      x[r][c] = plane[r][c] ^ (m->consecutive_xor ? plane[r - 1][c] : plane[0][c]);

  /* scan: (8L/16) x 16 array, untouched cells stay 0 */
  const int R = (BYTE * L) / 16;
  for (int r = 0; r < R; r++) scanned[r] = 0;
  for (int i = 0; i < m->table_size; i++) {
    if (x[m->rows[i]][m->cols[i]])
      scanned[i / 16] |= (uint16_t)(1u << (15 - (i % 16)));
  }
}

/* FPCModule::ProcessLine, FPCModule.cpp:19-85; sizes FPCModule.h:55
 * { ZRLE 7, Zero 4, SingleOne 7, TwoConsecOnes 8, FrontHalfZeros 12,
 *   BackHalfZeros 12, Uncompressible 17 }                                   */
int mpc_o_fpc_size(const uint16_t *scanned, int rows)
{
  int zrle = 0, run = 0, size = 0;
  for (int r = 0; r < rows; r++) {
    uint16_t v = scanned[r];
    if (v == 0) {
      if (zrle == 0) zrle = 1;
      run++;
      continue;
    }
    if (zrle == 1) {
      size += (run > 1) ? 7 : 4;
      zrle = 0;
      run = 0;
    }
    int ones = 0, pos[16];
    for (int c = 0; c < 16; c++)
      if (v & (1u << (15 - c))) pos[ones++] = c;
    if (ones <= 1)
      size += 7;                               /* isRowSingleOne */
    else if (ones == 2 && pos[1] - pos[0] == 1)
      size += 8;                               /* isRowTwoConsecOnes */
    else if ((v & 0xFF00u) == 0)
      size += 12;                              /* cols 0..7 zero */
    else if ((v & 0x00FFu) == 0)
      size += 12;                              /* cols 8..15 zero */
    else
      size += 17;
  }
  if (zrle == 1) size += (run > 1) ? 7 : 4;
  return size;
}

/* ------------------------------------------------------------------------ */
/* statistics: VPC.h:47-76, CompResult.h:30-35                              */
/* ------------------------------------------------------------------------ */

static void vpc_update(mpc_o_vpc_stats *st, unsigned uncomp, unsigned comp, int sel)
{
  if (!st) return;
  const int k = sel + 1;
  st->lines++;
  st->original_bits += uncomp;
  st->compressed_bits += comp;
  st->comp_ratio = (double)st->original_bits / (double)st->compressed_bits;
  st->c_original_bits[k] += uncomp;
  st->c_compressed_bits[k] += comp;
  st->c_comp_ratio[k] = (double)st->c_original_bits[k] / (double)st->c_compressed_bits[k];
  st->count[k]++;
  if (st->hist && comp < st->hist_bins)
    st->hist[(uint64_t)k * st->hist_bins + comp]++;
}

/* VPC::updateResidueStat, VPC.cpp:417-443; ResidueModule::GetMAE/GetMSE,
 * ResidueModule.cpp:43-74: residues over ALL positions as unsigned bytes.   */
static void vpc_update_residue(const mpc_o_config *cfg, const uint8_t *line, int sel,
                               mpc_o_vpc_stats *st)
{
  if (!st) return;
  const int L = cfg->line_size, k = sel + 1;
  double mae = 0, mse = 0;
  uint64_t sr = 0, sr2 = 0;
  if (sel != -1) {
    uint8_t pred[MPC_O_MAX_LINE];
    mpc_o_predict(&cfg->modules[sel], L, line, pred);
    for (int i = 0; i < L; i++) {
      uint8_t r = (uint8_t)(line[i] - pred[i]);
      mae += fabs((double)r);
      mse += pow((double)r, 2);
      sr += r;
      sr2 += (uint64_t)r * r;
    }
  } else {
    for (int i = 0; i < L; i++) {
      uint8_t r = line[i];
      mae += fabs((double)r);
      mse += pow((double)r, 2);
      sr += r;
      sr2 += (uint64_t)r * r;
    }
  }
  mae /= (double)L;
  mse /= (double)L;
  /* VPCResult::UpdateResidueStat, VPC.h:62-76 */
  st->sum_mae[k] += mae;
  st->sum_mse[k] += mse;
  st->residue_lines[k]++;
  st->mae[k] = st->sum_mae[k] / (double)st->residue_lines[k];
  st->mse[k] = st->sum_mse[k] / (double)st->residue_lines[k];
  st->sum_r[k] += sr;
  st->sum_r2[k] += sr2;
}

/* ------------------------------------------------------------------------ */
/* VPC::CompressLine, VPC.cpp:22-70, 332-415                                */
/* ------------------------------------------------------------------------ */

unsigned mpc_o_vpc_line(const mpc_o_config *cfg, const uint8_t *line, int *selected,
                        mpc_o_vpc_stats *st)
{
  const int L = cfg->line_size, M = cfg->num_modules;
  const unsigned uncomp = (unsigned)L * BYTE;
  const int has_aws = (M > 1 && cfg->modules[1].kind == MPC_O_ALLWORDSAME);

  /* checkAllZeros, VPC.cpp:332-347 + AllZeroModule.cpp:7-15 */
  {
    int zero = 1;
    for (int i = 0; i < L; i++)
      if (line[i] != 0) { zero = 0; break; }
    if (zero) {
      unsigned size = 0 + (unsigned)cfg->enc_bits[0 + 1];
      vpc_update(st, uncomp, size, 0);
      if (selected) *selected = 0;
      return size;
    }
  }
  /* checkAllWordSame, VPC.cpp:349-364 + AllWordSameModule.cpp:7-21 */
  if (has_aws) {
    int same = 1;
    for (int i = 4; i < L; i++)
      if (line[i] != line[i % 4]) { same = 0; break; }
    if (same) {
      unsigned size = 4 * BYTE + (unsigned)cfg->enc_bits[1 + 1];
      vpc_update(st, uncomp, size, 1);
      if (selected) *selected = 1;
      return size;
    }
  }
  /* checkOtherPatterns, VPC.cpp:366-415 */
  const int start = has_aws ? 2 : 1;
  const int R = (BYTE * L) / 16;
  int chosen = -1, best_z = 0, best_rows = 0;
  uint16_t best[MPC_O_MAX_TABLE / 16], cur[MPC_O_MAX_TABLE / 16];
  for (int i = start; i < M; i++) {
    mpc_o_scanned(&cfg->modules[i], L, line, cur);
    int z = 0;
    for (int r = 0; r < R; r++) {
      if (cur[r] == 0) z++; else break;
    }
    if (best_z <= z) { /* ties go to the later module */
      chosen = i;
      best_z = z;
      memcpy(best, cur, sizeof(uint16_t) * (size_t)R);
      best_rows = R;
    }
  }
  int enc = mpc_o_fpc_size(best, best_rows);
  unsigned size;
  if ((unsigned)enc < uncomp) {
    size = (unsigned)enc;
  } else {
    chosen = -1;
    size = uncomp;
  }
  size += (unsigned)cfg->enc_bits[chosen + 1];
  vpc_update(st, uncomp, size, chosen);
  vpc_update_residue(cfg, line, chosen, st);
  if (selected) *selected = chosen;
  return size;
}

void mpc_o_vpc_batch(const mpc_o_config *cfg, const uint8_t *lines, uint64_t n,
                     uint16_t *sizes, int8_t *selected, mpc_o_vpc_stats *st)
{
  const int L = cfg->line_size;
  for (uint64_t i = 0; i < n; i++) {
    int sel;
    unsigned s = mpc_o_vpc_line(cfg, lines + i * (uint64_t)L, &sel, st);
    if (sizes) sizes[i] = (uint16_t)s;
    if (selected) selected[i] = (int8_t)sel;
  }
}

/* ------------------------------------------------------------------------ */
/* BDI: BDI.cpp                                                             */
/* ------------------------------------------------------------------------ */

/* BDI::reduceSign, BDI.cpp:203-218 */
uint64_t mpc_o_bdi_reduce_sign(uint64_t x)
{
  uint64_t t = x >> 63;
  if (t) {
    for (int i = 62; i >= 0; i--) {
      t = (x >> i) & 1;
      if (t == 0)
        return x & (0xffffffffffffffffull >> (63 - (i + 1)));
    }
  }
  return x;
}

/* BDI::checkBDI, BDI.cpp:108-201.  The "sign extension" block (:150-151) is a
 * no-op (AND with all ones), so values are zero-extended little-endian.     */
unsigned mpc_o_bdi_check(const uint8_t *line, int L, unsigned base_size, unsigned delta_size)
{
  uint64_t limit = 0;
  switch (delta_size) {
  case 1: limit = 0xffull; break;
  case 2: limit = 0xffffull; break;
  case 4: limit = 0xffffffffull; break;
  }
  const unsigned n = (unsigned)L / base_size; /* maskSize */
  uint64_t v[MPC_O_MAX_LINE / 2];
  uint8_t mask[MPC_O_MAX_LINE / 2];
  for (unsigned i = 0; i < n; i++) {
    uint64_t t = 0;
    for (int j = (int)base_size - 1; j >= 0; j--)
      t = (t << BYTE) | line[i * base_size + (unsigned)j];
    v[i] = t;
    mask[i] = 0;
  }
  unsigned imm = 0;
  for (unsigned i = 0; i < n; i++) {
    if (mpc_o_bdi_reduce_sign(v[i]) <= limit) {
      mask[i] = 1;
      imm++;
    }
  }
  uint64_t base = 0;
  unsigned base_idx = 0;
  for (unsigned i = 0; i < n; i++) {
    if (!mask[i]) {
      base = v[i];
      base_idx = i;
      break;
    }
  }
  int not_all_delta = 0;
  for (unsigned i = base_idx + 1; i < n; i++) {
    if (!mask[i]) {
      if (mpc_o_bdi_reduce_sign(base - v[i]) > limit) {
        not_all_delta = 1;
        break;
      }
    }
  }
  /* 32-bit unsigned arithmetic, including the wrap when imm == n */
  if (not_all_delta)
    return n + BYTE * ((imm * delta_size) + ((n - imm) * base_size));
  else
    return n + BYTE * ((imm * delta_size) + (base_size + (n - imm - 1u) * delta_size));
}

/* BDI::CompressLine, BDI.cpp:6-74 */
unsigned mpc_o_bdi_line(const uint8_t *line, int L, int *selected, mpc_o_bdi_stats *st)
{
  const unsigned uncomp = BYTE * (unsigned)L;
  int select = 8; /* Uncompressed */
  unsigned best = uncomp, cur;

  int zeros = 1;
  for (int i = 0; i < L; i++)
    if (line[i] != 0) { zeros = 0; break; }
  int repeated = 1; /* isRepeated(line, 8), BDI.cpp:84-106 */
  for (int i = 8; i < (L / 8) * 8; i++)
    if (line[i] != line[i % 8]) { repeated = 0; break; }

  if (zeros) {
    best = BYTE;
    select = 0;
  } else if (repeated) {
    best = BYTE * 8;
    select = 1;
  } else {
    static const unsigned combos[6][2] = {{8, 1}, {8, 2}, {8, 4}, {4, 1}, {4, 2}, {2, 1}};
    for (int k = 0; k < 6; k++) {
      cur = mpc_o_bdi_check(line, L, combos[k][0], combos[k][1]);
      if (best > cur) {
        select = 2 + k;
        best = cur;
      }
    }
    if (best == uncomp) select = 8;
  }
  unsigned size = best + 4;
  if (st) {
    st->lines++;
    st->original_bits += uncomp;
    st->compressed_bits += size;
    st->comp_ratio = (double)st->original_bits / (double)st->compressed_bits;
    st->counts[select]++;
  }
  if (selected) *selected = select;
  return size;
}

void mpc_o_bdi_batch(const uint8_t *lines, int L, uint64_t n, uint16_t *sizes,
                     int8_t *selected, mpc_o_bdi_stats *st)
{
  for (uint64_t i = 0; i < n; i++) {
    int sel;
    unsigned s = mpc_o_bdi_line(lines + i * (uint64_t)L, L, &sel, st);
    if (sizes) sizes[i] = (uint16_t)s;
    if (selected) selected[i] = (int8_t)sel;
  }
}

/* ---------------------------------------------------------------------------
 * FPC (reference src/compressor/FPC.cpp:7-88, FPC.h:31-38).  Words are little
 * endian (concatenate(), FPC.cpp:90-108); the tests run in the reference's order.
 * ------------------------------------------------------------------------- */
static int fpc_prefix(uint32_t val)
{
  if (val == 0u) return 0;
  if ((val & 0xFFFFFFF8u) == 0u || (val & 0xFFFFFFF8u) == 0xFFFFFFF8u) return 1;      /* 4-bit sign-extended   */
  if ((val & 0xFFFFFF80u) == 0u || (val & 0xFFFFFF80u) == 0xFFFFFF80u) return 2;      /* 8-bit sign-extended   */
  if ((val & 0xFFFF8000u) == 0u || (val & 0xFFFF8000u) == 0xFFFF8000u) return 3;      /* 16-bit sign-extended  */
  if ((val & 0x0000FFFFu) == 0u) return 4;                                            /* halfword + zero pad   */
  {
    const uint32_t m = val & 0xFF80FF80u;                                             /* two byte-sized halves */
    if (m == 0u || m == 0xFF800000u || m == 0x0000FF80u || m == 0xFF80FF80u) return 5;
  }
  if ((val & 0xFFu) == ((val >> 8) & 0xFFu) && (val & 0xFFu) == ((val >> 16) & 0xFFu) &&
      (val & 0xFFu) == ((val >> 24) & 0xFFu)) return 6;                               /* repeated bytes        */
  return 7;
}

unsigned mpc_o_fpc_line(const uint8_t *line, int L, mpc_o_fpc_stats *st)
{
  static const unsigned kBits[8] = {3 + 3, 4 + 3, 8 + 3, 16 + 3, 16 + 3, 16 + 3, 8 + 3, 32 + 3};   /* + PREFIX_SIZE */
  const int n = L / 4;
  unsigned size = 0;
  int i = 0;
  while (i < n) {
    const uint32_t val = (uint32_t)line[4 * i] | ((uint32_t)line[4 * i + 1] << 8) | ((uint32_t)line[4 * i + 2] << 16) |
                         ((uint32_t)line[4 * i + 3] << 24);
    const int p = fpc_prefix(val);
    unsigned bits = kBits[p];
    if (p == 0 && i > 0) {
      /* words after the first of a zero run cost nothing (FPC.cpp:26-30) */
      const uint32_t prev = (uint32_t)line[4 * i - 4] | ((uint32_t)line[4 * i - 3] << 8) |
                            ((uint32_t)line[4 * i - 2] << 16) | ((uint32_t)line[4 * i - 1] << 24);
      if (prev == 0u) bits = 0;
    }
    size += bits;
    if (st) {
      st->original_bits += 32;
      st->compressed_bits += bits;
      st->total_words++;
      st->counts[p]++;
    }
    i++;
  }
  if (st) {
    st->lines++;
    st->comp_ratio = (double)st->original_bits / (double)st->compressed_bits;
  }
  return size;
}

void mpc_o_fpc_batch(const uint8_t *lines, int L, uint64_t n, uint16_t *sizes, mpc_o_fpc_stats *st)
{
  for (uint64_t i = 0; i < n; i++) {
    const unsigned s = mpc_o_fpc_line(lines + i * (uint64_t)L, L, st);
    if (sizes) sizes[i] = (uint16_t)s;
  }
}

/* ---------------------------------------------------------------------------
 * BPC (reference src/compressor/BPC.cpp:20-185, BPC.h:12-33)
 * ------------------------------------------------------------------------- */
static void bpc_pattern(mpc_o_bpc_stats *st, unsigned num_words, int selected)
{
  if (!st) return;
  st->total_words += num_words;      /* BPC.h:29-33 */
  st->counts[selected]++;
}

unsigned mpc_o_bpc_line(const uint8_t *line, int L, mpc_o_bpc_stats *st)
{
  static const unsigned kZrl[34] = {0, 3, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7,
                                    7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7};   /* BPC.cpp:7 */
  const int n = L / 4;
  int64_t data[MPC_O_MAX_LINE / 4], deltas[MPC_O_MAX_LINE / 4];
  for (int i = 0; i < n; i++)      /* zero-extended words (see mpc_oracle.h) */
    data[i] = (int64_t)((uint32_t)line[4 * i] | ((uint32_t)line[4 * i + 1] << 8) | ((uint32_t)line[4 * i + 2] << 16) |
                        ((uint32_t)line[4 * i + 3] << 24));
  const int nd = n - 1;
  for (int r = 1; r < n; r++) deltas[r - 1] = data[r] - data[r - 1];
  int32_t DBP[33], DBX[33], prev = 0;
  for (int col = 32; col >= 0; col--) {
    int32_t buf = 0;
    for (int row = nd - 1; row >= 0; row--) buf = (int32_t)(((uint32_t)buf << 1) | (uint32_t)((deltas[row] >> col) & 1));
    DBP[col] = buf;
    DBX[col] = col == 32 ? buf : (buf ^ prev);
    prev = buf;
  }
  unsigned length = 3 + 4;                 /* encodeFirst: always the 4-bit sign-extended case */
  unsigned run = 0;
  for (int i = 32; i >= 0; i--) {
    if (DBX[i] == 0) {
      run++;
      continue;
    }
    if (run > 0) {
      length += kZrl[run];
      bpc_pattern(st, run, 1);             /* ZRLE */
    }
    run = 0;
    if (DBP[i] == 0) {
      length += 5;
      bpc_pattern(st, 1, 2);               /* Zero */
    } else if (DBX[i] == 0x7fffffff) {
      length += 5;
      bpc_pattern(st, 1, 6);               /* AllOnes */
    } else {
      const uint32_t x = (uint32_t)DBX[i];
      const int ones = __builtin_popcount(x);
      if (ones == 1) {
        length += 10;
        bpc_pattern(st, 1, 3);             /* SingleOne */
      } else if (ones == 2 && (x & (x >> 1)) != 0) {
        length += 10;
        bpc_pattern(st, 1, 4);             /* ConsecTwoOnes */
      } else {
        length += 32;
        bpc_pattern(st, 1, 0);             /* Uncomp */
      }
    }
  }
  if (run > 0) {
    length += kZrl[run];
    bpc_pattern(st, run, 1);
  }
  if (st) {
    st->lines++;
    st->original_bits += 8ull * (uint64_t)L;
    st->compressed_bits += length;
    st->comp_ratio = (double)st->original_bits / (double)st->compressed_bits;
  }
  return length;
}

void mpc_o_bpc_batch(const uint8_t *lines, int L, uint64_t n, uint16_t *sizes, mpc_o_bpc_stats *st)
{
  for (uint64_t i = 0; i < n; i++) {
    const unsigned s = mpc_o_bpc_line(lines + i * (uint64_t)L, L, st);
    if (sizes) sizes[i] = (uint16_t)s;
  }
}

