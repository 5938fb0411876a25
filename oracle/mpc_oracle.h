/*
 * mpc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's per-cache-line evaluator
 * (scalable-arch/CAL_22-MPC: VPC multi-prediction path + BDI baseline).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path
 * (cal_22-mpc_amd/csrc) never links, loads or calls anything in oracle/.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - stages a-5,a-6,a-8,a-9,a-11 (predictors, residue, XOR, scan, common
 *     encoder) are checked against the reference's own translation units
 *     compiled unmodified into oracle/_ref (tests/test_oracle_vs_ref.py, runs
 *     where /root/reference exists) and against committed vectors made by
 *     that build (tests/golden/ref_stage_vectors.json);
 *   - whole-line VPC sizes / BDI sizes are checked against the known answers
 *     SURVEY.md 8c captured from the reference object code;
 *   - BDI.cpp, VPC.cpp, AllZero/AllWordSame/Bitplane modules cannot be built
 *     here (they pull in un-vendored strutil.h/json/json.h): those rows are
 *     restated from source and pinned by the SURVEY known answers only.
 *
 * All citations are relative to /root/reference/src/compressor/.
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_O_MAX_LINE 256
#define MPC_O_MAX_MODULES 32
#define MPC_O_MAX_TABLE (8 * MPC_O_MAX_LINE)

/* module kinds (VPC.cpp:125-320) */
enum {
  MPC_O_ALLZERO = 0,
  MPC_O_ALLWORDSAME = 1,
  MPC_O_PREDCOMP = 2
};

/* predictor kinds (VPCmodules/PredictorModule.cpp) */
enum {
  MPC_O_PRED_WEIGHT = 0,
  MPC_O_PRED_DIFF = 1,
  MPC_O_PRED_ONEBASE = 2,
  MPC_O_PRED_CONSEC = 3
};

typedef struct {
  int32_t kind;             /* MPC_O_ALLZERO / ALLWORDSAME / PREDCOMP */
  int32_t pred_kind;        /* MPC_O_PRED_* (PREDCOMP only) */
  int32_t root;             /* RootIndex */
  int32_t consecutive_xor;  /* XORModule.consecutiveXOR */
  int32_t table_size;       /* ScanModule.TableSize */
  int32_t base[MPC_O_MAX_LINE];    /* BaseIndexTable */
  float   weight[MPC_O_MAX_LINE];  /* WeightTable (as parsed, float) */
  int32_t diff[MPC_O_MAX_LINE];    /* DiffTable */
  int32_t rows[MPC_O_MAX_TABLE];   /* ScanModule.Rows */
  int32_t cols[MPC_O_MAX_TABLE];   /* ScanModule.Cols */
} mpc_o_module;

typedef struct {
  int32_t num_modules;                     /* overview.num_modules (M) */
  int32_t line_size;                       /* overview.lineSize (L) */
  int32_t enc_bits[MPC_O_MAX_MODULES + 1]; /* index k -> cluster k-1 */
  mpc_o_module modules[MPC_O_MAX_MODULES];
} mpc_o_config;

/* Statistics, following VPC.h:16-76 (ClusterStat / VPCResult) and
 * CompResult.h:30-35.  Cluster index k = cluster + 1.                       */
typedef struct {
  uint64_t lines;
  uint64_t original_bits;
  uint64_t compressed_bits;
  double   comp_ratio;
  /* per cluster */
  uint64_t count[MPC_O_MAX_MODULES + 1];
  uint64_t c_original_bits[MPC_O_MAX_MODULES + 1];
  uint64_t c_compressed_bits[MPC_O_MAX_MODULES + 1];
  double   c_comp_ratio[MPC_O_MAX_MODULES + 1];
  /* residue statistics: the reference's running doubles (VPC.h:62-76) ... */
  double   sum_mae[MPC_O_MAX_MODULES + 1];
  double   sum_mse[MPC_O_MAX_MODULES + 1];
  double   mae[MPC_O_MAX_MODULES + 1];
  double   mse[MPC_O_MAX_MODULES + 1];
  uint64_t residue_lines[MPC_O_MAX_MODULES + 1];
  /* ... and the same quantities as exact integer sums (what the HIP path
   * accumulates; equal to the doubles while L is a power of two).          */
  uint64_t sum_r[MPC_O_MAX_MODULES + 1];
  uint64_t sum_r2[MPC_O_MAX_MODULES + 1];
  /* histogram[k][size]; hist_bins entries per cluster, caller allocated */
  uint64_t *hist;
  uint32_t hist_bins;
} mpc_o_vpc_stats;

typedef struct {
  uint64_t lines;
  uint64_t original_bits;
  uint64_t compressed_bits;
  double   comp_ratio;
  uint64_t counts[9];   /* BDI.h:10-21 BDIState order */
} mpc_o_bdi_stats;

/* Returns 0 if the configuration is one the reference can run without
 * undefined behaviour (layout rule of VPC.cpp:312-376, table bounds), else a
 * negative code.                                                            */
int mpc_o_vpc_validate(const mpc_o_config *cfg);

/* One line through VPC::CompressLine (VPC.cpp:22-70).  `line` has
 * cfg->line_size bytes.  Returns the compressed size in bits, writes the
 * selected cluster (-1 .. M-1) to *selected, updates *st if non-NULL.       */
unsigned mpc_o_vpc_line(const mpc_o_config *cfg, const uint8_t *line,
                        int *selected, mpc_o_vpc_stats *st);

/* n lines, contiguous; sizes/selected may be NULL. */
void mpc_o_vpc_batch(const mpc_o_config *cfg, const uint8_t *lines, uint64_t n,
                     uint16_t *sizes, int8_t *selected, mpc_o_vpc_stats *st);

/* Stage-level entry points (used to compare against oracle/_ref). */
void mpc_o_predict(const mpc_o_module *m, int L, const uint8_t *line, uint8_t *pred);
void mpc_o_residue(const mpc_o_module *m, int L, const uint8_t *line, uint8_t *res);
/* scanned: R = 8L/16 rows, 16 bits each, bit (15-c) of scanned[r] = column c */
void mpc_o_scanned(const mpc_o_module *m, int L, const uint8_t *line, uint16_t *scanned);
int  mpc_o_fpc_size(const uint16_t *scanned, int rows);

/* BDI::CompressLine (BDI.cpp:6-74). selected = BDIState value. */
unsigned mpc_o_bdi_line(const uint8_t *line, int L, int *selected, mpc_o_bdi_stats *st);
void mpc_o_bdi_batch(const uint8_t *lines, int L, uint64_t n, uint16_t *sizes,
                     int8_t *selected, mpc_o_bdi_stats *st);
uint64_t mpc_o_bdi_reduce_sign(uint64_t x);
unsigned mpc_o_bdi_check(const uint8_t *line, int L, unsigned base_size, unsigned delta_size);

/* FPC::CompressLine (FPC.cpp:7-88), frequent pattern compression of the line's L/4
 * little-endian 32-bit words.  PARITY UNPINNED: the reference ships no fixture for it and
 * FPC.cpp does not compile here without the un-vendored strutil.h, so this follows the
 * source text only.  One deliberate definition: the zero-run loop of the reference
 * (`while(dataConcat[i] == 0)`, FPC.cpp:26) reads past the end of the line when the run
 * reaches it (undefined behaviour, the result depends on heap contents); here a run ends
 * at the end of the line.                                                          */
typedef struct {
  uint64_t lines;
  uint64_t original_bits;     /* 32 per word (FPC.h:31-38) */
  uint64_t compressed_bits;
  double   comp_ratio;
  uint64_t total_words;
  uint64_t counts[8];         /* FPCState Prefix0..7 */
} mpc_o_fpc_stats;
unsigned mpc_o_fpc_line(const uint8_t *line, int L, mpc_o_fpc_stats *st);

/* BPC::CompressLine (BPC.cpp:20-185), bit-plane compression: deltas of consecutive 32-bit
 * words, 33 delta bit planes (DBP), each XORed with the plane above (DBX), run-length /
 * pattern coded.  PARITY UNPINNED like FPC (no fixture; BPC.cpp needs the un-vendored
 * strutil.h to compile).  Two properties of the source are kept as they are:
 *   - encodeFirst (BPC.cpp:96-108) tests `if (base = 0)`, an assignment, so it always
 *     returns 3 + 4 = 7 bits;
 *   - pattern ZeroDBP (5) is never counted, a zero DBP is counted as Zero (2).
 * One definition where the source has undefined behaviour: a word is copied into the low
 * 4 bytes of an uninitialised int64_t (BPC.cpp:42-44); here the upper half is zero (the word
 * is zero-extended), which makes every delta a 33-bit two's complement number.          */
typedef struct {
  uint64_t lines;
  uint64_t original_bits;
  uint64_t compressed_bits;
  double   comp_ratio;
  uint64_t total_words;       /* BPCResult::TotalWords: sum of UpdatePattern's numWords */
  uint64_t counts[7];         /* BPCPattern order */
} mpc_o_bpc_stats;
unsigned mpc_o_bpc_line(const uint8_t *line, int L, mpc_o_bpc_stats *st);
void mpc_o_bpc_batch(const uint8_t *lines, int L, uint64_t n, uint16_t *sizes, mpc_o_bpc_stats *st);
void mpc_o_fpc_batch(const uint8_t *lines, int L, uint64_t n, uint16_t *sizes, mpc_o_fpc_stats *st);

#ifdef __cplusplus
}
#endif
#endif
