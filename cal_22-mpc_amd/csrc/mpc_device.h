// mpc_device.h -- parameter blocks shared by the host side (mpc_capi.hip) and the
// gfx950 kernels (mpc_kernels.hip).  Plain C structs, passed by value as kernel
// arguments.
#pragma once

#ifndef __HIPCC_RTC__
#include <stdint.h>
#else      /* hiprtc (mpc_jit.h): no <stdint.h>; the fixed-width names the kernels use */
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long uintptr_t;
#endif

#define MPC_MAX_MODULES 32        /* M (AllZero + AllWordSame + prediction modules) */
#define MPC_MAX_PRED    31        /* prediction (PredComp) modules */
#define MPC_MAX_LINE    256       /* bytes per line, generic path */

/* fast-path predictor forms (see DESIGN.md "Fast path") */
enum {
  MPC_FK_ONEBASE = 0,  /* every byte predicted by byte 0                         */
  MPC_FK_CONSEC  = 1,  /* previous byte of the byte-plane-shuffled line          */
  MPC_FK_DIFF    = 2,  /* base byte in own/previous dword (v_perm) + constant     */
  MPC_FK_WEIGHT  = 3,  /* base byte in own/previous dword (v_perm); shift class 1 is the
                          unshifted one (ls1 = rs1 = 0), class 2 any single shift           */
  MPC_FK_WEIGHT2 = 4   /* same with two shifted classes                                   */
};

struct MpcFastModule {
  int32_t kind;       /* MPC_FK_* */
  int32_t cx;         /* consecutiveXOR */
  int32_t ls1, rs1;   /* WEIGHT: first shift class  ((b << ls1) >> rs1) & c1 */
  int32_t ls2, rs2;   /* WEIGHT: second shift class ((b << ls2) >> rs2) & c2 */
  int32_t tab_off;    /* DIFF/WEIGHT: dword offset of {sel, c1, c2, c1 & 0x7f.., c1 & 0x80..}[L/4 each] in tab */
  int32_t root;       /* RootIndex (0 on the plain unrolled kernels, 0..15 on their general-layout twins; any position for
                         ONEBASE / DIFF / WEIGHT* on the run-time loop) */
  int32_t gather;     /* DIFF / WEIGHT*: 1 = the BaseIndexTable is not windowed (a base byte outside the own / previous dword): the
                         base bytes are gathered with the table as compile-time constants (gather_unrolled configurations) */
  int32_t wgen;       /* WEIGHT*: 1 = the WeightTable has more than two shift distances: the predicted word is assembled from the
                         table as compile-time constants (wshift_unrolled configurations) */
  uint32_t plane_mask; /* this module's scanned bit planes, in every byte (planes_differ configurations; else unused) */
  int32_t prev_word;  /* DIFF/WEIGHT periodic tables.  1: every base byte of words 1.. is the same byte of the
                         previous word (BaseIndexTable[i] = i - 4) and the table entries of words 1.. are
                         identical: no v_perm_b32, two table entries instead of L/4.  2: the same two words
                         back (i - 8, 8-byte elements) from word 2 on, entries repeating every 2 words.  0: no */
};

/* generic-path module tables: byte offsets into MpcVpcParams::gtab */
struct MpcGenModule {
  int32_t pred_kind;  /* 0 weight, 1 diff, 2 onebase, 3 consecutive */
  int32_t root;
  int32_t cx;
  int32_t table_size;
  int32_t off_base;   /* uint8 [L]  BaseIndexTable                  */
  int32_t off_shift;  /* int8  [L]  weight shift distance (clamped) */
  int32_t off_diff;   /* uint8 [L]  (uint8_t)DiffTable              */
  int32_t off_rows;   /* uint8 [table_size]                         */
  int32_t off_cols;   /* uint8 [table_size]                         */
  int32_t off_scan;   /* uint16 [table_size]  Cols | Rows << 8 (4-byte aligned): one read per scanned bit */
  int32_t pad[2];
};

struct MpcVpcParams {
  int32_t L;            /* line size in bytes */
  int32_t M;            /* num_modules */
  int32_t n_pred;       /* number of PredComp modules */
  int32_t start;        /* module index of the first PredComp module (1 or 2) */
  int32_t has_aws;      /* module 1 is AllWordSame */
  int32_t hist_bins;    /* bins per cluster */
  int32_t runtime_only; /* fast path, but only through the run-time module loop: a RootIndex above 15, the byte-major order, a
                           scan table of fewer than 16 entries */
  int32_t tab_words;    /* number of dwords in tab */
  int32_t trunc_off;    /* truncated plane-major scan table (TableSize < 8 L, the same for every module): dword offset in
                           tab of L/4 mask words (the bits of the XORed residue bytes that are scanned); -1: full table */
  int32_t byte_major;   /* scan order of every module: 0 plane-major (scanned bit i = plane i / L of byte i % L), 1 byte-major
                           (bit i = plane i % 8 of byte i / 8: row r of the scanned array = residue bytes 2r, 2r+1) */
  int32_t enc_bits[MPC_MAX_MODULES + 1];  /* index cluster+1 */
  struct MpcFastModule fm[MPC_MAX_PRED];
  struct MpcGenModule gm[MPC_MAX_PRED];
  const uint32_t *tab;  /* fast path dword tables (device) */
  const uint8_t *gtab;  /* generic path byte tables (device) */
  int32_t gtab_bytes;   /* size of gtab; the generic kernel keeps a copy in LDS when it fits */
  int32_t gen_layout;   /* 1: some RootIndex is 1..15 and / or the plane-major scan table is truncated (16 <= TableSize < 8 L):
                           the unrolled kernels' general-layout twins take it */
  uint32_t plane_mask;  /* the scanned bit planes of a residue byte, in every byte (0xffffffff: all; TableSize = 6 L: 0xfcfcfcfc);
                           0xffffffff also for a table cut inside a plane, whose per-word masks sit at trunc_off */
  int32_t bm_unrolled;  /* 1: byte-major order with every RootIndex 0 and a complete first row (TableSize >= 16): no built-in
                           kernel (runtime_only stays 1), but the unrolled kernels can be compiled for it at handle creation
                           (mpc_jit.h, -DMPC_JIT_BM); gen_layout then says whether the table is truncated */
  int32_t anyroot_unrolled; /* 1: plane-major order, complete first row, some RootIndex above 15: no built-in kernel (runtime_only
                           stays 1), but the general-layout kernels can be compiled for it at handle creation with the roots as
                           constants (mpc_jit.h, -DMPC_JIT_ANYROOT); gen_layout is 1 */
  int32_t wshift_unrolled; /* 1: some WEIGHT module has more than two shift distances (like gather_unrolled: -DMPC_JIT_WSHIFT) */
  int32_t gather_unrolled; /* 1: some DIFF / WEIGHT module has a BaseIndexTable that is not windowed: only a kernel compiled at
                           creation with the tables as constants (mpc_jit.h, -DMPC_JIT_GATHER) evaluates it on the fast path;
                           without one the handle falls back to the generic kernel */
  int32_t planes_differ; /* 1: plane-major tables that stop after a whole number of bit planes, NOT the same number in every
                           module (MpcFastModule::plane_mask): only a kernel compiled at creation evaluates it (mpc_jit.h,
                           -DMPC_JIT_PLANES); without one the handle falls back to the generic kernel */
};

/* Device-side raw statistics (uint64 each):
 *   VPC: [0,K) sum_r   [K,2K) sum_r2   [2K, 2K + K*bins) histogram[k][size]
 *   BDI: [0,9) Counts  [9] compressed_bits                                   */
static inline uint64_t mpc_vpc_raw_len(int K, int bins) { return 2ull * (uint64_t)K + (uint64_t)K * (uint64_t)bins; }
#define MPC_BDI_RAW_LEN 10
/*   FPC: [0,8) Counts (Prefix0..7)  [8] compressed_bits                        */
#define MPC_FPC_RAW_LEN 9
/*   BPC: [0,7) Counts (BPCPattern order)  [7] TotalWords  [8] compressed_bits  */
#define MPC_BPC_RAW_LEN 9
