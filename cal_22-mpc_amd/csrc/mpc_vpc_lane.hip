// mpc_vpc_lane.hip -- the hot kernel: VPC multi-prediction evaluation of lines whose
// configuration has the plane-major scan, RootIndex 0 and "windowed" predictor tables
// (see mpc_config.h: build_vpc_plan); 32 / 64 / 128 byte lines.  gfx950 / wave64 only.
//
// Mapping: one LANE per line.  Lane i of a wave holds a whole line in W = L/4 registers.  The
// unrolled-sequence kernels bring groups of 64 lines from global memory straight into a per-wave
// ring in LDS (LDS-DMA, mpc_ring.h) and read them out one line per lane; the run-time module loop
// loads its lines with L/16 global_load_dwordx4 per lane.  Nothing
// crosses lanes: the per-line predicates are per-lane booleans, a __ballot of one gives
// the wave-uniform decisions (skip the prediction modules of an all-zero wave, skip a
// module that cannot win anywhere in the wave, skip the encoder when every line is
// certified incompressible).  An earlier kernel of this round held a line in L/16 lanes;
// the byte work per line was the same SWAR arithmetic, but everything that is per line --
// masks, selector, winner bookkeeping, statistics, loop control -- was issued once per 16
// lines instead of once per 64, and the kernel is issue-bound.
//
// What the stages compute, against the reference (src/compressor/...):
//   AllZero / AllWordSame      VPCmodules/AllZeroModule.cpp:7-15, AllWordSameModule.cpp:7-21
//   predictors                 VPCmodules/PredictorModule.cpp:37-173
//   residue (root first)       VPCmodules/ResidueModule.cpp:12-41
//   bit-plane + XOR            BitplaneModule.cpp:7-51, XORModule.cpp:5-23; on bytes:
//                              g = b ^ (b >> 1) (consecutive) or b ^ (msb ? 0x7f : 0), column 0 untouched
//   selector (leading zero rows, ties -> later module)   VPC.cpp:366-395
//   common encoder             VPCmodules/FPCModule.cpp:19-85 (sizes FPCModule.h:55)
//   decision + id bits         VPC.cpp:397-407
//   residue statistics         VPC.cpp:417-443, ResidueModule.cpp:43-74
//
// The kernel is a template over the sequence of predictor forms so that the module
// sequence is unrolled; configurations whose sequence has no instantiation run the
// same stages with a run-time loop over the modules.
#include "mpc_kernel_common.h"
#include "mpc_ring.h"

// 1 only in a translation unit that mpc_jit.h hands to the run-time compiler for a configuration with the BYTE-MAJOR scan
// order (scanned bit i = plane i % 8 of byte i / 8: row r of the scanned array = XORed residue bytes 2r, 2r+1): the
// unrolled kernels then use that order's selector, row-0 prefilter (bytes 0 and 1 all zero), certificate and encoder.
// No built-in kernel is compiled this way.
#ifndef MPC_JIT_BM
#define MPC_JIT_BM 0
#endif
// 1 only in a translation unit that mpc_jit.h compiles for a configuration with a RootIndex above 15: MPC_JIT_ROOTS then
// lists every module's root, the group code sees them as constants, and the general form of the root-first rotation (any
// word, the prefilter corrected for a root outside row 0) folds to a few instructions -- with roots read at run time the
// same code costs 33 spilled VGPRs.
#ifndef MPC_JIT_ANYROOT
#define MPC_JIT_ANYROOT 0
#endif
// 1 only in a translation unit that mpc_jit.h compiles for a configuration whose plane-major tables stop after a whole number
// of bit planes that DIFFERS between the modules: the selector takes each module's own mask (MpcFastModule::plane_mask) and the
// winner's mask travels with it to the XOR stage.
#ifndef MPC_JIT_PLANES
#define MPC_JIT_PLANES 0
#endif
// 1 only in a translation unit that mpc_jit.h compiles for a configuration in which a DiffBase / WeightBase module's
// BaseIndexTable is not windowed: MPC_JIT_BASES then holds every module's table (-1: predicted byte forced to 0),
// MPC_JIT_GATHERS which modules gather, and the base bytes of a word come from wherever the table says -- constants, so each
// word costs at most three v_perm_b32.
#ifndef MPC_JIT_GATHER
#define MPC_JIT_GATHER 0
#endif
// 1 only in a translation unit that mpc_jit.h compiles for a configuration in which a WeightBase module's WeightTable has more
// than two shift distances: MPC_JIT_SHIFTS holds every module's shift per byte (99: the predicted byte is 0), MPC_JIT_WGENS
// which modules take them from there.
#ifndef MPC_JIT_WSHIFT
#define MPC_JIT_WSHIFT 0
#endif
#ifndef MPC_ABLATE
#define MPC_ABLATE 0   // development-only timing ablations (tools/ablate.sh); results are WRONG when non-zero
#endif
namespace {

// lane-kernel predictor kinds: MPC_FK_* plus LK_PW for periodic tables (MpcFastModule::prev_word):
// every base byte of words 1.. is the same byte of the previous word and the table entries of
// words 1.. are identical -- no v_perm_b32, two table entries (word 0, the other words)
constexpr int LK_PW = 8;
constexpr int LK_PW2 = 16;    // the same two words back (prev_word == 2: BaseIndexTable[i] = i - 8)
__device__ __host__ constexpr int lk_base(int kind) { return kind & 7; }
__device__ __host__ constexpr int lk_dist(int kind) { return (kind & LK_PW2) ? 2 : ((kind & LK_PW) ? 1 : 0); }

// The predictor tables are read through the CONSTANT address space: wave-uniform reads of it are scalar loads
// (s_load into SGPRs) whatever else the kernel does.  Read as ordinary global memory they turn into vector loads as
// soon as the compiler sees anything it must take for a store in front of them (the line ring's asm statements),
// and a vector load's s_waitcnt vmcnt(0) also waits for every line request in flight.
typedef const u32 __attribute__((address_space(4))) *ctab_t;
__device__ __forceinline__ ctab_t lane_tab(const MpcVpcParams &P) { return (ctab_t)(u64)P.tab; }

#if MPC_JIT_ANYROOT || MPC_JIT_GATHER || MPC_JIT_WSHIFT
template <int Q>
__device__ __forceinline__ MpcFastModule lane_fm_const(const MpcVpcParams &P)
{
  MpcFastModule f = P.fm[Q];
#if MPC_JIT_ANYROOT
  constexpr int roots[] = {MPC_JIT_ROOTS};
  f.root = roots[Q];
#endif
#if MPC_JIT_GATHER
  constexpr int gathers[] = {MPC_JIT_GATHERS};
  f.gather = gathers[Q] ? Q + 1 : 0;      // (a constant: which table lane_gather_base reads)
#endif
#if MPC_JIT_WSHIFT
  constexpr int wgens[] = {MPC_JIT_WGENS};
  f.wgen = wgens[Q] ? Q + 1 : 0;
#endif
  return f;
}
#define LANE_FM(Q) lane_fm_const<Q>(P)
#else
#define LANE_FM(Q) P.fm[Q]
#endif

template <int W>
struct Lane {
  u32 x[W];    // the line
  u32 b0;      // byte 0 = root (RootIndex 0)
  u32 rootb;   // root replicated into 4 bytes
  bool gen;    // a constant of the instantiation: general layout (a RootIndex of 0..15 per module, truncated plane-major tables)
  ctab_t trunc;  // general layout, a table cut inside a bit plane: per residue word the bits that are scanned (else null)
};

// ---- general layout (the GEN twins of the unrolled kernels) ---------------------------------------------------------
// RootIndex 1..15 (OneBase / DiffBase / WeightBase; wave-uniform, read from the module's parameters).  The predictors
// work on the natural byte positions and leave the RAW root byte in the natural residue (the tables force the predicted
// byte to 0 there); ResidueModule.cpp:24-39 then moves the root to the front, i.e. bytes 0..root rotate by one position.
// With the root inside the first 16 bytes that rotation stays inside residue words 0..3 = row 0 of every bit plane, so
//  * the row-0 prefilters need no rotation at all (they OR the MSBs of bytes 0..15: the same set of bytes),
//  * a full evaluation rotates four words (two instructions each).
// (Any root -- the word picked among all W, a wave-uniform test per rotated word, the prefilter corrected for a root
// outside row 0 -- was measured too: 33 spilled VGPRs and 334 spilled SGPRs in the four-module twin.  Roots above 15
// stay on the run-time loop.)
// line[root] in every byte
template <int W>
__device__ __forceinline__ u32 lane_root_bytes(const Lane<W> &c, int root)
{
  // OR of masked words: a select chain here is turned into an indexed read of a copy of x[] in scratch memory
  const int rw = root >> 2;
  u32 xw = 0;
#pragma unroll
  for (int e = 0; e < (MPC_JIT_ANYROOT ? W : 4); e++) xw = and_or(c.x[e], 0u - (u32)(e == rw), xw);     // (a constant root: one word)
  return perm(xw, xw, 0x01010101u * (u32)(root & 3));
}
// what OneBase predicts for word e: line[root] everywhere, 0 at the root position itself
__device__ __forceinline__ u32 lane_onebase_pred(u32 rb, int root, int e)
{
  const u32 here = (e == (root >> 2)) ? (0xffu << (8 * (root & 3))) : 0u;      // (scalar)
  return rb & ~here;
}
// row-0 prefilter with a root above 15 (MPC_JIT_ANYROOT): row 0 is the raw root and natural bytes 0..14 -- natural byte 15 has
// moved to row 1 and does not count
__device__ __forceinline__ u32 lane_row0_word(u32 msbs, int root, int e)
{
  return (MPC_JIT_ANYROOT && e == 3 && root > 15) ? (msbs & 0x00ffffffu) : msbs;
}
// natural residue -> root-first order, words 0..NR-1 (run-time roots: NR <= 4 suffices, root <= 15)
template <int W, int NR>
__device__ __forceinline__ void lane_root_to_front(const Lane<W> &c, int root, u32 (&r)[W])
{
  u32 prev = lane_root_bytes<W>(c, root) & 0xff000000u;      // the raw root enters byte 0 of word 0
#pragma unroll
  for (int e = 0; e < (MPC_JIT_ANYROOT ? NR : (NR < 4 ? NR : 4)); e++) {
    const u32 sh = alignbyte(r[e], prev, 3);                   // the natural bytes one position up
    const int nb = root + 1 - 4 * e;                           // bytes of this word at positions <= root
    const u32 m = nb >= 4 ? ~0u : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
    prev = r[e];
    r[e] = mask_sel(m, sh, r[e]);
  }
}

// bit 7 of every byte = MSB of the per-byte difference a - b (the other bits are junk)
__device__ __forceinline__ u32 msb_of_bsub(u32 a, u32 b) { return ((a | H80) - (b & L7F)) ^ a ^ ~b; }

// byte k of the four words x[4q..4q+3] (one row of their 4x4 byte transpose)
template <int W>
__device__ __forceinline__ u32 plane_bytes(const u32 (&x)[W], int q, int k)
{
  const u32 s01 = k >= 2 ? 0x07030602u : 0x05010400u;
  const u32 p01 = perm(x[4 * q + 1], x[4 * q], s01), p23 = perm(x[4 * q + 3], x[4 * q + 2], s01);
  return perm(p23, p01, (k & 1) ? 0x07060302u : 0x05040100u);
}

// dword j of the byte-plane-shuffled line `inp` (PredictorModule.cpp:143-155):
// inp[n] = byte (3 - n / W) of word n % W
template <int W>
__device__ __forceinline__ u32 shuffled_word(const u32 (&x)[W], int j)
{
  constexpr int NG = W / 4;
  return plane_bytes<W>(x, j % NG, 3 - j / NG);
}

// predicted word from the base bytes b: b + diff constant, or b shifted / masked per weight class
template <int KIND>
__device__ __forceinline__ u32 window_predict(u32 b, u32 c1, u32 c2, const MpcFastModule &fm)
{
  if constexpr (KIND == MPC_FK_DIFF)
    return ((b & L7F) + c1) ^ ((b & H80) ^ c2);   // per-byte b + diff; the host split diff into c1 = low 7 bits, c2 = MSBs
  else if constexpr (KIND == MPC_FK_WEIGHT)
    return (b & c1) | ((b >> fm.rs2) & c2);
  else
    return (((b << fm.ls1) >> fm.rs1) & c1) | (((b << fm.ls2) >> fm.rs2) & c2);
}

#if MPC_JIT_GATHER
// base bytes of word e of module q from anywhere in the line (q, e and the table are constants once this is inlined)
template <int W>
__device__ __forceinline__ u32 lane_gather_base(const Lane<W> &c, int q, int e)
{
  constexpr short bases[][4 * W] = {MPC_JIT_BASES};
  u32 half[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int b0 = bases[q][4 * e + 2 * h], b1 = bases[q][4 * e + 2 * h + 1];
    const u32 w0 = b0 >= 0 ? c.x[b0 >> 2] : 0u, w1 = b1 >= 0 ? c.x[b1 >> 2] : 0u;
    const u32 sel = (b0 >= 0 ? (u32)(b0 & 3) : 0x0cu) | ((b1 >= 0 ? 4u + (u32)(b1 & 3) : 0x0cu) << 8) | 0x0c0c0000u;
    half[h] = perm(w1, w0, sel);            // [byte b0, byte b1, 0, 0]
  }
  return perm(half[1], half[0], 0x05040100u);
}
#endif

#if MPC_JIT_WSHIFT
// WeightBase with any number of shift distances: the predicted word from the base bytes b, byte by byte (PredictorModule.cpp:57-62:
// base >> |s| or base << s in uint8); q, e and the table are constants once this is inlined, equal shifts fold into one
template <int W>
__device__ __forceinline__ u32 lane_weight_general(u32 b, int q, int e)
{
  constexpr signed char sh[][4 * W] = {MPC_JIT_SHIFTS};
  u32 pred = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int s = sh[q][4 * e + k];
    if (s == 99) continue;
    const u32 keep = (s >= 0 ? ((0xffu << s) & 0xffu) : (0xffu >> -s)) << (8 * k);      // the bits that stay inside the byte
    pred |= (s >= 0 ? (b << s) : (b >> -s)) & keep;
  }
  return pred;
}
#endif

// base bytes of word e: own / previous word through v_perm_b32, or simply the previous word
template <int W, int KIND>
__device__ __forceinline__ u32 window_base(const Lane<W> &c, int e, ctab_t t)
{
  constexpr int D = lk_dist(KIND);
  if (D && e >= D) return c.x[e - D];
  return perm(c.x[e], e ? c.x[e - 1] : 0u, t[e]);
}

// residue word e of a DiffBase / WeightBase module.  full = false: only bit 7 of each
// byte is meaningful (the MSBs the row-0 prefilter looks at)
template <int W, int KIND, bool FULL = true>
__device__ __forceinline__ u32 window_residue(const Lane<W> &c, int e, ctab_t t, const MpcFastModule &fm)
{
#if MPC_JIT_GATHER
  const u32 b = fm.gather ? lane_gather_base<W>(c, fm.gather - 1, e) : window_base<W, KIND>(c, e, t);
#else
  const u32 b = window_base<W, KIND>(c, e, t);
#endif
  constexpr int D = lk_dist(KIND);
  const int k = D == 2 ? (e < 2 ? e : 2 + (e & 1)) : (D == 1 ? (e ? 1 : 0) : e);     // table entry of word e
  if constexpr (lk_base(KIND) == MPC_FK_WEIGHT) {
    // class 1 unshifted (mask c1), class 2 shifted right (mask c2, MSBs clear): the low 7 bits
    // and the MSB of the predicted byte come from the pre-split masks t[3W..], t[4W..]
    const u32 p7 = ((b >> fm.rs2) & t[2 * W + k]) | (b & t[3 * W + k]);
    const u32 sub = (c.x[e] | H80) - p7;
    const u32 w = c.x[e] ^ (b & t[4 * W + k]);          // bit 7: line ^ predicted
    return FULL ? (sub ^ (~w & H80)) : (sub ^ ~w);
  } else {
#if MPC_JIT_WSHIFT
    const u32 pred = fm.wgen ? lane_weight_general<W>(b, fm.wgen - 1, e) : window_predict<lk_base(KIND)>(b, t[W + k], t[2 * W + k], fm);
#else
    const u32 pred = window_predict<lk_base(KIND)>(b, t[W + k], t[2 * W + k], fm);
#endif
    return FULL ? bsub(c.x[e], pred) : msb_of_bsub(c.x[e], pred);
  }
}

// Residue bytes (root first == natural order for RootIndex 0) of one module;
// root_r = the residue GetMAE / GetMSE see at the root position.  NR: only words 0..NR-1.
template <int W, int KIND, int NR = W>
__device__ __forceinline__ void lane_residue(const Lane<W> &c, const MpcFastModule &fm, ctab_t tab,
                                             u32 (&r)[W], u32 &root_r)
{
  root_r = 0;
  if constexpr (lk_base(KIND) == MPC_FK_ONEBASE) {
    if (c.gen && fm.root != 0) {       // (wave-uniform; GEN twins only)
      const u32 rb = lane_root_bytes<W>(c, fm.root);
#pragma unroll
      for (int e = 0; e < NR; e++) r[e] = bsub(c.x[e], (MPC_JIT_ANYROOT || e < 4) ? lane_onebase_pred(rb, fm.root, e) : rb);
    } else {
      // predicted = line[0] everywhere; position 0 of the residue array is the raw root
      r[0] = bsub(c.x[0], c.rootb & 0xffffff00u);
#pragma unroll
      for (int e = 1; e < NR; e++) r[e] = bsub(c.x[e], c.rootb);
    }
  } else if constexpr (lk_base(KIND) == MPC_FK_CONSEC) {
    // predicted[i] = inp[i - 1]; predicted byte 0 := 0 keeps the raw root at position 0
    u32 in[NR];
#pragma unroll
    for (int j = 0; j < NR; j++) in[j] = shuffled_word<W>(c.x, j);
    r[0] = bsub(c.x[0], in[0] << 8);
#pragma unroll
    for (int e = 1; e < NR; e++) r[e] = bsub(c.x[e], alignbyte(in[e], in[e - 1], 3));
    // predicted[root] = inp[0] = byte 3 of word 0 (PredictorModule.cpp:159-164)
    root_r = (c.b0 - (c.x[0] >> 24)) & 0xffu;
  } else {
    // windowed tables; they force the predicted root byte to 0, so residue[0] = raw root
    ctab_t t = tab + fm.tab_off;
#pragma unroll
    for (int e = 0; e < NR; e++) r[e] = window_residue<W, KIND>(c, e, t, fm);
  }
}

// Row-0 prefilter: a module other than the last one can only win the selector with at
// least one leading zero row (ties go to the later module, VPC.cpp:389), i.e. only if
// the MSBs of residue bytes 0..15 are all clear.  Returns the MSBs of residue words
// E0..E1-1 ORed together (bit 7 of each byte).
template <int W, int KIND, int E0, int E1>
__device__ __forceinline__ u32 lane_row0(const Lane<W> &c, const MpcFastModule &fm, ctab_t tab)
{
  u32 m = 0;
  if constexpr (lk_base(KIND) == MPC_FK_ONEBASE) {
    if (c.gen && fm.root != 0) {       // (wave-uniform; GEN twins only)
      const u32 rb = lane_root_bytes<W>(c, fm.root);
#pragma unroll
      for (int e = E0; e < E1; e++) m |= lane_row0_word(msb_of_bsub(c.x[e], lane_onebase_pred(rb, fm.root, e)), fm.root, e);
      if (MPC_JIT_ANYROOT && E0 == 0 && fm.root > 15) m |= rb;            // the raw root sits in row 0
    } else {
#pragma unroll
      for (int e = E0; e < E1; e++) m |= msb_of_bsub(c.x[e], e ? c.rootb : (c.rootb & 0xffffff00u));
    }
  } else if constexpr (lk_base(KIND) == MPC_FK_CONSEC) {
#pragma unroll
    for (int e = E0; e < E1; e++) {
      const u32 in_e = shuffled_word<W>(c.x, e);
      m |= msb_of_bsub(c.x[e], e ? alignbyte(in_e, shuffled_word<W>(c.x, e - 1), 3) : (in_e << 8));
    }
  } else {
    ctab_t t = tab + fm.tab_off;
#pragma unroll
    for (int e = E0; e < E1; e++) m |= lane_row0_word(window_residue<W, KIND, false>(c, e, t, fm), c.gen ? fm.root : 0, e);
    if (MPC_JIT_ANYROOT && c.gen && E0 == 0 && fm.root > 15) m |= lane_root_bytes<W>(c, fm.root);     // the raw root sits in row 0
  }
  return m & H80;
}

// Leading zero rows of the scanned array (row = plane * W/4 + column group).  The first
// non-zero row of the XORed planes equals that of the raw residue planes (DESIGN.md
// "Selector on raw residues"), so the selector works on r directly.
// plane_mask: the scanned bit planes, in every byte (GEN twins: a table that stops after whole planes; else all ones)
template <int W>
__device__ __forceinline__ u32 lane_leading_zero_rows(const u32 (&r)[W], ctab_t scan_mask = nullptr, u32 plane_mask = ~0u)
{
  constexpr int NG = W / 4;
  u32 S[NG], G = 0;
  if (scan_mask) {
    // truncated scan table (run-time loop only): bits that are not scanned do not count
#pragma unroll
    for (int j = 0; j < NG; j++) {
      S[j] = (r[4 * j] & scan_mask[4 * j]) | (r[4 * j + 1] & scan_mask[4 * j + 1]) | (r[4 * j + 2] & scan_mask[4 * j + 2]) |
             (r[4 * j + 3] & scan_mask[4 * j + 3]);
      G |= S[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < NG; j++) {
      S[j] = r[4 * j] | r[4 * j + 1] | r[4 * j + 2] | r[4 * j + 3];
      G |= S[j];
    }
  }
  const u32 gf = fold8(G) & plane_mask;              // (& 0xff of the mask: fold8 leaves one byte)
  const u32 p = (u32)__clz((int)gf) - 24u;           // first non-zero scanned plane (0 = MSB); 8 if none
  const u32 B = H80 >> (p & 7u);                      // that plane's bit in every byte
  u32 j_star = NG - 1;
#pragma unroll
  for (int j = NG - 2; j >= 0; j--) j_star = (S[j] & B) ? (u32)j : j_star;
  return (plane_mask == ~0u ? G : gf) ? NG * p + j_star : 2u * W;
}

template <int W>
struct LaneBest {
  u32 r[W];
  u32 z;
  int q;
  u32 root_r;
  u32 cx;
  u32 pm;      // MPC_JIT_PLANES: the winner's scanned bit planes
  u32 encb;    // id bits of the winner's cluster (wave-uniform scalar loads, selected per lane: an
               // indexed read of the kernel arguments per lane would be a vector load + full wait)
};

// ---- module sequence -------------------------------------------------------------
// Pass 1: row-0 prefilters of all modules but the last -> wave-uniform bit mask of the
// modules that have to be evaluated (bit Q: enough needed lines have a clear row 0).
//
// Deferral.  A module that only a few lines of the group can win with would still be evaluated
// for all 64 lanes.  When at most MPC_DEFER_MAX lines pass a module's prefilter (and the wave's
// queue has room), those LINES are set aside instead -- their index goes to the wave's queue in
// LDS and they take no further part in this group -- and the module is skipped for the group.
// Queued lines are evaluated later, 64 at a time, with every module they need (drain groups:
// no deferral there).  The result of a line does not depend on which way it went: the prefilter
// only ever skips modules that cannot win.
#ifndef MPC_DEFER_MAX
#define MPC_DEFER_MAX 8
#endif
// (queue entries per wave: LaneEnv::defer_cap, chosen by the host with the LDS plan; the streaming loop hands over
// to a drain above defer_cap - 64 entries)

// Paired groups.  A wave with a two-stage ring holds 128 consecutive lines (a block) and evaluates them as two groups
// of 64.  Normally the groups are the first and the second 64 lines.  When the prefilters show that neighbouring
// lines alternate between two kinds (arrays of 128-byte records; the interleaved integer / floating-point trace of
// BASELINE config 4) the wave switches to PAIRED groups -- the even lines of the block, then the odd ones (lane i:
// lines 2i and 2i+1) -- so that a group holds ONE kind of line and pays only for the modules that kind needs.  The
// lines come out of LDS either way, so the choice costs no memory traffic (round 2 paid 16 % extra HBM traffic for
// it with strided register loads); the wave still probes with plain groups every kPairProbe blocks and stays with
// them unless the sign is back, because the read-out of a paired group has 2-way LDS bank conflicts and the next
// block can only be requested after both groups have been read.
#ifndef MPC_PAIR_PROBE
#define MPC_PAIR_PROBE 64
#endif
constexpr u32 kPairProbe = MPC_PAIR_PROBE;

// LDS bytes of the statistics: as vpc_stats_smem plus the spare histogram slot deferred lines are parked in
__host__ __device__ static inline size_t lane_stats_smem(int K, int bins)
{
  return 16 * (size_t)((2 * K * 8 + 15) / 16) + 16 * (size_t)(((K * bins + 2) * 4 + 15) / 16);    // (+ the workgroup's group counter)
}

// First stage of pass 1 for all modules at once: the smallest of the modules' row-0 MSB words (residue words
// 0..2) is non-zero exactly when every module shows an MSB; if that holds on every needed line, no module
// but the last can win anywhere in the group -- one ballot instead of one per module (the common case on
// incompressible data).
// byte-major order (MPC_JIT_BM; RootIndex 0): row 0 of the scanned array is residue bytes 0 and 1 -- a module other than the
// last can only win if both are zero.  Returns them (non-zero: the module cannot win on this line).
template <int W, int KIND>
__device__ __forceinline__ u32 lane_row0_bm(const Lane<W> &c, const MpcFastModule &fm, ctab_t tab)
{
  u32 r[W], root_r;
  lane_residue<W, KIND, 1>(c, fm, tab, r, root_r);
  return r[0] & (c.trunc ? (c.trunc[0] & 0xffffu) : 0xffffu);
}

template <int W, int NPT, int Q>
__device__ __forceinline__ u32 lane_row0_min(const Lane<W> &, const MpcVpcParams &) { return ~0u; }

template <int W, int NPT, int Q, int KIND, int... REST>
__device__ __forceinline__ u32 lane_row0_min(const Lane<W> &c, const MpcVpcParams &P)
{
  if constexpr (Q + 1 == NPT) {
    return ~0u;
  } else {
    const u32 m = MPC_JIT_BM ? lane_row0_bm<W, KIND>(c, LANE_FM(Q), lane_tab(P)) : lane_row0<W, KIND, 0, 3>(c, LANE_FM(Q), lane_tab(P));
    return min(m, lane_row0_min<W, NPT, Q + 1, REST...>(c, P));
  }
}

template <int W, int NPT, int Q>
__device__ __forceinline__ u32 lane_prefilters(const Lane<W> &, const MpcVpcParams &, u64 &, u64 &, bool, bool &) { return 0; }

// alt: set when a kept module is passed by lines of one parity only (at least 16 of them) while lines of the other
// parity need the modules too -- the sign of a trace whose neighbouring lines alternate between two kinds (see
// the paragraph on paired groups)
template <int W, int NPT, int Q, int KIND, int... REST>
__device__ __forceinline__ u32 lane_prefilters(const Lane<W> &c, const MpcVpcParams &P, u64 &need_mask, u64 &defer_mask,
                                               bool allow_defer, bool &alt)
{
  if constexpr (Q + 1 == NPT) {
    return 0;
  } else {
    u32 bit = 0;
    if constexpr (!(MPC_ABLATE & 8)) {
      // words 0..2 first: on incompressible data 12 bytes almost always show an MSB
      u32 msb = MPC_JIT_BM ? lane_row0_bm<W, KIND>(c, LANE_FM(Q), lane_tab(P)) : lane_row0<W, KIND, 0, 3>(c, LANE_FM(Q), lane_tab(P));
      if ((__ballot(msb != 0) & need_mask) != need_mask) {
        if (!MPC_JIT_BM) msb |= lane_row0<W, KIND, 3, 4>(c, LANE_FM(Q), lane_tab(P));
        const u64 pass = ~__ballot(msb != 0) & need_mask;
        if (pass) {
          if (MPC_DEFER_MAX > 0 && W <= 16 && allow_defer && __popcll(pass) <= MPC_DEFER_MAX) {   // (128-byte lines: measured slower)
            defer_mask |= pass;
            need_mask &= ~pass;
          } else {
            bit = 1u << Q;
            const u64 even = 0x5555555555555555ull;
            if (__popcll(pass) >= 16 && ((pass & even) == 0 || (pass & ~even) == 0) &&
                __popcll(need_mask & ~pass) >= 16)
              alt = true;
          }
        }
      }
    }
    return bit | lane_prefilters<W, NPT, Q + 1, REST...>(c, P, need_mask, defer_mask, allow_defer, alt);
  }
}

template <int W, int NR>
__device__ __forceinline__ u32 lane_leading_zero_rows_bm(const u32 (&r)[W], ctab_t scan_mask, u32 none);

// Pass 2 (only when pass 1 kept a module): the kept modules and the last one in order,
// winner updated with bit masks (v_bitop3_b32).
template <int W, int NPT, int Q>
__device__ __forceinline__ void lane_seq(const Lane<W> &, const MpcVpcParams &, u32, LaneBest<W> &, bool) {}

template <int W, int NPT, int Q, int KIND, int... REST>
__device__ __forceinline__ void lane_seq(const Lane<W> &c, const MpcVpcParams &P, u32 keep_bits, LaneBest<W> &best,
                                         bool any_full)
{
  constexpr bool last = Q + 1 == NPT;
  if (last || (keep_bits & (1u << Q))) {
    u32 r[W], root_r;
    lane_residue<W, KIND>(c, LANE_FM(Q), lane_tab(P), r, root_r);
    if (c.gen && LANE_FM(Q).root != 0) lane_root_to_front<W, W>(c, LANE_FM(Q).root, r);
    const u32 pm_q = MPC_JIT_PLANES ? LANE_FM(Q).plane_mask : (c.gen ? P.plane_mask : ~0u);
    const u32 z = MPC_JIT_BM ? lane_leading_zero_rows_bm<W, W>(r, c.gen ? c.trunc : nullptr, 2u * W)
                             : lane_leading_zero_rows<W>(r, c.gen ? c.trunc : nullptr, pm_q);
    if (!any_full) {
      // the first module evaluated for the group (wave-uniform): it is the winner so far, nothing to select
      best.z = z;
      best.q = Q;
      best.root_r = root_r;
      best.cx = (u32)LANE_FM(Q).cx;
      best.pm = pm_q;
      best.encb = (u32)P.enc_bits[P.start + Q + 1];
#pragma unroll
      for (int e = 0; e < W; e++) best.r[e] = r[e];
    } else {
      // ties go to the later module (VPC.cpp:389)
      const u32 m = best.z <= z ? ~0u : 0u;
      best.z = mask_sel(m, z, best.z);
      best.q = (int)mask_sel(m, (u32)Q, (u32)best.q);
      best.root_r = mask_sel(m, root_r, best.root_r);
      best.cx = mask_sel(m, (u32)LANE_FM(Q).cx, best.cx);
      if (MPC_JIT_PLANES) best.pm = mask_sel(m, pm_q, best.pm);
      best.encb = mask_sel(m, (u32)P.enc_bits[P.start + Q + 1], best.encb);
#pragma unroll
      for (int e = 0; e < W; e++) best.r[e] = mask_sel(m, r[e], best.r[e]);
    }
    any_full = true;
  }
  if constexpr (!last) lane_seq<W, NPT, Q + 1, REST...>(c, P, keep_bits, best, any_full);
}

// the last module alone (pass 1 ruled out every other module for the whole wave)
template <int W, int NPT, int Q>
__device__ __forceinline__ void lane_last(const Lane<W> &, const MpcVpcParams &, LaneBest<W> &) {}

template <int W, int NPT, int Q, int KIND, int... REST>
__device__ __forceinline__ void lane_last(const Lane<W> &c, const MpcVpcParams &P, LaneBest<W> &best)
{
  if constexpr (Q + 1 == NPT) {
    lane_residue<W, KIND>(c, LANE_FM(Q), lane_tab(P), best.r, best.root_r);
    if (c.gen && LANE_FM(Q).root != 0) lane_root_to_front<W, W>(c, LANE_FM(Q).root, best.r);
    best.q = Q;
    best.cx = (u32)LANE_FM(Q).cx;
    best.pm = MPC_JIT_PLANES ? LANE_FM(Q).plane_mask : ~0u;
    best.encb = (u32)P.enc_bits[P.start + Q + 1];
  } else {
    lane_last<W, NPT, Q + 1, REST...>(c, P, best);
  }
}

// ---- byte-major scan order (run-time module loop only) -----------------------------------------
// scanned bit i = plane i % 8 of byte i / 8 (ScanModule.cpp:13-19 with Rows[i] = i % 8, Cols[i] = i / 8): row r
// of the scanned array is the pair of XORed residue bytes (2r, 2r+1), most significant bit first.  A byte of
// the XORed residue is zero exactly when the raw residue byte is (b ^ (b >> 1) and b ^ (msb ? 0x7f : 0) are
// bijections that fix 0; the same holds for the leading bits kept by a truncated table), so the selector again
// works on raw residues.

// leading zero rows = (index of the first non-zero scanned byte) / 2; NR words looked at, none non-zero: `none`
template <int W, int NR>
__device__ __forceinline__ u32 lane_leading_zero_rows_bm(const u32 (&r)[W], ctab_t scan_mask, u32 none)
{
  u32 fe = NR, fw = 0;
#pragma unroll
  for (int e = NR - 1; e >= 0; e--) {
    const u32 v = scan_mask ? (r[e] & scan_mask[e]) : r[e];
    fe = v ? (u32)e : fe;
    fw = v ? v : fw;
  }
  const u32 byte = 4u * fe + ((u32)__builtin_ctz(fw | 0x80000000u) >> 3);      // little-endian: the lowest set bit's byte
  return fe == (u32)NR ? none : (byte >> 1);
}

// common encoder (FPCModule.cpp:19-85) over the 2 W rows of a byte-major scanned array
template <int W>
__device__ __forceinline__ u32 lane_encode_bm(const u32 (&t)[W])
{
  u32 bits = 0;
  u64 Z = 0;                       // bit r: row r is zero
#pragma unroll
  for (int e = 0; e < W; e++) {
    const u32 hw = perm(t[e], t[e], 0x02030001u);       // bytes swapped inside the halves: half = (first byte << 8) | second byte
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const u32 h = k ? (hw >> 16) : (hw & 0xffffu);    // the row, column 0 = bit 15
      const u32 ones = (u32)__popc(h);
      const bool adj = (h & (h >> 1)) != 0u;
      const bool half_empty = (h & 0xff00u) == 0u || (h & 0x00ffu) == 0u;
      const u32 cost = ones == 1u ? 7u : (ones == 2u && adj) ? 8u : half_empty ? 12u : 17u;
      bits += h ? cost : 0u;
      Z |= (u64)(h == 0u ? 1u : 0u) << (2 * e + k);
    }
  }
  // zero rows are coded as runs in row order: 4 bits per run, 7 if longer than one row
  const u64 all = W == 32 ? ~0ull : ((1ull << (2 * (W & 31))) - 1ull);
  const u64 starts = Z & ~(Z << 1);
  const u64 longer = starts & (Z >> 1) & all;
  return bits + 4u * (u32)__popcll(starts) + 3u * (u32)__popcll(longer);
}

// word `idx` (wave-uniform) of the line
template <int W>
__device__ __forceinline__ u32 lane_word_at(const u32 (&x)[W], int idx)
{
  // OR of masked words: a select chain here is turned into an indexed read of a copy of x[] in scratch memory
  u32 w = 0;
#pragma unroll
  for (int e = 0; e < W; e++) w = and_or(x[e], 0u - (u32)(e == idx), w);
  return w;
}

// run-time module sequence (configurations whose sequence has no instantiation, a RootIndex other
// than 0, a truncated scan table).
//
// Order.  The last module is evaluated first, the others follow in DESCENDING order and replace the
// winner only with strictly more leading zero rows -- the choice of the reference's ascending loop
// with `<=` (ties go to the later module, VPC.cpp:377-395).  Before a module is evaluated in full its
// residue words 0..3 (column group 0) give an upper bound of its leading zero rows: a bit in plane p
// of group 0 means at most (L/16) * p of them.  When that cannot beat the winner so far on any line of
// the group the module is skipped (on incompressible data: every module but the last).
//
// RootIndex != 0 (OneBase / DiffBase / WeightBase): the predictors work on the natural byte
// positions -- the tables force the predicted byte at the root position to 0, so the natural
// residue holds the raw root there -- and ResidueModule.cpp:24-39 then moves the root to the front:
// bytes 0..root of the natural residue rotate by one position, the rest stays.
template <int W, int NR>
__device__ __forceinline__ void lane_residue_rt(const Lane<W> &c, const MpcVpcParams &P, const MpcFastModule &fm, u32 (&r)[W],
                                                u32 &root_r)
{
  switch (fm.kind) {
  case MPC_FK_ONEBASE:
    if (fm.root == 0) {
      lane_residue<W, MPC_FK_ONEBASE, NR>(c, fm, lane_tab(P), r, root_r);
    } else {
      const int rw = fm.root >> 2, rk = fm.root & 3;
      const u32 xw = lane_word_at<W>(c.x, rw);
      const u32 rb = ((xw >> (8 * rk)) & 0xffu) * 0x01010101u;     // line[root] in every byte
      const u32 here = 0xffu << (8 * rk);
#pragma unroll
      for (int e = 0; e < NR; e++) r[e] = bsub(c.x[e], e == rw ? (rb & ~here) : rb);   // predicted 0 at the root: raw root
      root_r = 0;
    }
    break;
  case MPC_FK_CONSEC: lane_residue<W, MPC_FK_CONSEC, NR>(c, fm, lane_tab(P), r, root_r); break;
  case MPC_FK_DIFF:
    if (fm.prev_word == 2) lane_residue<W, MPC_FK_DIFF | LK_PW2, NR>(c, fm, lane_tab(P), r, root_r);
    else if (fm.prev_word == 1) lane_residue<W, MPC_FK_DIFF | LK_PW, NR>(c, fm, lane_tab(P), r, root_r);
    else lane_residue<W, MPC_FK_DIFF, NR>(c, fm, lane_tab(P), r, root_r);
    break;
  case MPC_FK_WEIGHT:
    if (fm.prev_word == 2) lane_residue<W, MPC_FK_WEIGHT | LK_PW2, NR>(c, fm, lane_tab(P), r, root_r);
    else if (fm.prev_word == 1) lane_residue<W, MPC_FK_WEIGHT | LK_PW, NR>(c, fm, lane_tab(P), r, root_r);
    else lane_residue<W, MPC_FK_WEIGHT, NR>(c, fm, lane_tab(P), r, root_r);
    break;
  default:
    if (fm.prev_word == 2) lane_residue<W, MPC_FK_WEIGHT2 | LK_PW2, NR>(c, fm, lane_tab(P), r, root_r);
    else if (fm.prev_word == 1) lane_residue<W, MPC_FK_WEIGHT2 | LK_PW, NR>(c, fm, lane_tab(P), r, root_r);
    else lane_residue<W, MPC_FK_WEIGHT2, NR>(c, fm, lane_tab(P), r, root_r);
    break;
  }
  if (fm.root != 0) {
    // root to the front: residue[0] = natural[root] (the raw root byte), residue[j] = natural[j - 1] for 1 <= j <= root
    const int rw = fm.root >> 2, rk = fm.root & 3;
    const u32 rootraw = (lane_word_at<W>(c.x, rw) >> (8 * rk)) & 0xffu;
    u32 prev = rootraw << 24;              // what enters byte 0 of word 0
#pragma unroll
    for (int e = 0; e < NR; e++) {
      const u32 sh = alignbyte(r[e], prev, 3);                     // the natural bytes one position up
      const int nb = fm.root + 1 - 4 * e;                          // bytes of this word at positions <= root
      const u32 m = nb >= 4 ? ~0u : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
      prev = r[e];
      r[e] = mask_sel(m, sh, r[e]);
    }
  }
}

template <int W>
__device__ __forceinline__ void lane_seq_runtime(const Lane<W> &c, const MpcVpcParams &P, LaneBest<W> &best,
                                                 ctab_t scan_mask, u64 need_mask)
{
  constexpr u32 NG = W / 4;
  best.z = 0;
  best.q = -1;
  best.root_r = 0;
  best.cx = 0;
  best.encb = 0;
#pragma unroll
  for (int e = 0; e < W; e++) best.r[e] = 0;
  for (int q = P.n_pred - 1; q >= 0; q--) {
    const MpcFastModule fm = P.fm[q];
    const bool last = q == P.n_pred - 1;
    u32 r[W], root_r;
    if (!last) {
      // upper bound of this module's leading zero rows from column group 0
      lane_residue_rt<W, 4>(c, P, fm, r, root_r);
      bool may_win;
      if (P.byte_major) {
        // byte-major rows: a non-zero byte among the first 16 bounds the leading zero rows directly
        may_win = lane_leading_zero_rows_bm<W, 4>(r, scan_mask, ~0u) > best.z;
      } else {
        u32 g = scan_mask ? ((r[0] & scan_mask[0]) | (r[1] & scan_mask[1]) | (r[2] & scan_mask[2]) | (r[3] & scan_mask[3]))
                          : (r[0] | r[1] | r[2] | r[3]);
        const u32 p0 = (u32)__clz((int)fold8(g)) - 24u;                // first plane with a bit; 8 if none
        may_win = g == 0u || NG * p0 > best.z;
      }
      if ((__ballot(may_win) & need_mask) == 0) continue;            // wave-uniform
    }
    lane_residue_rt<W, W>(c, P, fm, r, root_r);
    const u32 z = P.byte_major ? lane_leading_zero_rows_bm<W, W>(r, scan_mask, 2u * W) : lane_leading_zero_rows<W>(r, scan_mask);
    if (last) {       // (wave-uniform) the first module evaluated: the winner so far
      best.z = z;
      best.q = q;
      best.root_r = root_r;
      best.cx = (u32)fm.cx;
      best.encb = (u32)P.enc_bits[P.start + q + 1];
#pragma unroll
      for (int e = 0; e < W; e++) best.r[e] = r[e];
    } else {
      const u32 m = z > best.z ? ~0u : 0u;
      best.z = mask_sel(m, z, best.z);
      best.q = (int)mask_sel(m, (u32)q, (u32)best.q);
      best.root_r = mask_sel(m, root_r, best.root_r);
      best.cx = mask_sel(m, (u32)fm.cx, best.cx);
      best.encb = mask_sel(m, (u32)P.enc_bits[P.start + q + 1], best.encb);
#pragma unroll
      for (int e = 0; e < W; e++) best.r[e] = mask_sel(m, r[e], best.r[e]);
    }
  }
}

// byte j of the result = OR of the four bytes of X[j] (NG == 4), or bytes 0 / 2 for X[0] / X[1] (NG == 2)
template <int NG>
__device__ __forceinline__ u32 fold_pack(const u32 (&X)[NG])
{
  const u32 a01 = perm(X[1], X[0], 0x05040100u) | perm(X[1], X[0], 0x07060302u);   // [X0.b0|b2, X0.b1|b3, X1.b0|b2, X1.b1|b3]
  if constexpr (NG == 2) {
    return (a01 | (a01 >> 8)) & 0x00ff00ffu;
  } else {
    const u32 a23 = perm(X[3], X[2], 0x05040100u) | perm(X[3], X[2], 0x07060302u);
    return perm(a23, a01, 0x06040200u) | perm(a23, a01, 0x07050301u);
  }
}

// Incompressibility certificate, a lower bound of the common encoder's size: a row whose
// two 8-column halves are both non-zero and that has a bit outside columns 7/8 costs 17
// bits (it is neither a single one nor an adjacent pair, FPCModule.cpp:50-69); any other
// non-zero row costs at least 7; a zero row sits in a run that costs at least 4.
// True: the encoding cannot beat 8*L bits, the encoder need not run.
template <int W>
__device__ __forceinline__ bool lane_certified(const u32 (&t)[W])
{
  constexpr int NG = W / 4;
  constexpr int GP = NG < 4 ? NG : 4;      // column groups per packed word
  u32 n17 = 0, nnz = 0;
#pragma unroll
  for (int h = 0; h < NG / GP; h++) {
    const u32 *th = &t[16 * h];
    // columns 0..6 and 9..15 of every group folded to plane masks; columns 7 and 8 are whole
    // bytes (byte 3 of word 1, byte 0 of word 2) and only need gathering
    u32 F[GP], B[GP];
#pragma unroll
    for (int j = 0; j < GP; j++) {
      F[j] = th[4 * j] | (th[4 * j + 1] & 0x00ffffffu);
      B[j] = (th[4 * j + 2] & 0xffffff00u) | th[4 * j + 3];
    }
    const u32 Sf7 = fold_pack<GP>(F), Sb7 = fold_pack<GP>(B);
    u32 C7, C8;
    if constexpr (GP == 2) {
      C7 = perm(th[5], th[1], 0x0c070c03u);
      C8 = perm(th[6], th[2], 0x0c040c00u);
    } else {
      C7 = perm(perm(th[13], th[9], 0x0c0c0703u), perm(th[5], th[1], 0x0c0c0703u), 0x05040100u);
      C8 = perm(perm(th[14], th[10], 0x0c0c0400u), perm(th[6], th[2], 0x0c0c0400u), 0x05040100u);
    }
    const u32 Sf = Sf7 | C7, Sb = Sb7 | C8;     // front / back half non-zero
    const u32 So = Sf7 | Sb7;                   // a bit outside columns 7 / 8
    n17 += (u32)__popc(Sf & Sb & So);
    nnz += (u32)__popc(Sf | Sb);
  }
  const u32 bound = 10u * n17 + 7u * nnz + (nnz != 2u * W ? 4u : 0u);
  return bound >= 32u * W;
}

// The same certificate for the byte-major order (MPC_JIT_BM): row r = bytes 2r (columns 0..7) and 2r+1 (columns 8..15) of t,
// two rows per word.  Packed 16-bit minima do the per-row tests: min(a, b) is non-zero exactly when both are.
typedef unsigned short lane_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 lane_pk_min(u32 a, u32 b)
{
  return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(lane_u16x2, a), __builtin_bit_cast(lane_u16x2, b)));
}
__device__ __forceinline__ u32 lane_pk_add(u32 a, u32 b)
{
  return __builtin_bit_cast(u32, (lane_u16x2)(__builtin_bit_cast(lane_u16x2, a) + __builtin_bit_cast(lane_u16x2, b)));
}
template <int W>
__device__ __forceinline__ bool lane_certified_bm(const u32 (&t)[W])
{
  u32 acc = 0;      // per halfword: rows that cost 17 in bits 8.., non-zero rows in bits 0..7
#pragma unroll
  for (int e = 0; e < W; e++) {
    const u32 x = t[e];
    const u32 front = x & 0x00ff00ffu, back = (x >> 8) & 0x00ff00ffu;      // columns 0..7 / 8..15 of the word's two rows
    const u32 both = lane_pk_min(front, back);                              // both halves non-zero
    const u32 c17 = lane_pk_min(both, x & 0x7ffe7ffeu);                     // ... and a bit outside columns 7 / 8
    // (flags 0 / 1 per row: a 17-bit row counts in bits 8.., a non-zero row in bits 0..7)
    acc = lane_pk_add(acc, (lane_pk_min(c17, 0x00010001u) << 8) | lane_pk_min(x, 0x00010001u));
  }
  const u32 sum = (acc & 0xffffu) + (acc >> 16);
  const u32 n17 = sum >> 8, nnz = sum & 0xffu;
  const u32 bound = 10u * n17 + 7u * nnz + (nnz != 2u * W ? 4u : 0u);
  return bound >= 32u * W;
}

// Common encoder's row classes for FOUR column groups at once (t = their 16 words): the 16
// columns are brought into "packed" form -- P[c] holds column c of the four groups, one
// byte per group, bit 7-p = plane p -- with 4x4 byte transposes, and every step of
// encode_rows (mpc_kernel_common.h) then runs once on packed words instead of once per
// group.  Returns the bits of the non-zero rows; Z = packed zero-row masks.
__device__ __forceinline__ u32 encode_rows_x4(const u32 *t, u32 &Z)
{
  u32 P[16];
#pragma unroll
  for (int e = 0; e < 4; e++) {     // word e of each group -> columns 4e .. 4e+3
    const u32 lo01 = perm(t[4 + e], t[e], 0x05010400u), hi01 = perm(t[4 + e], t[e], 0x07030602u);
    const u32 lo23 = perm(t[12 + e], t[8 + e], 0x05010400u), hi23 = perm(t[12 + e], t[8 + e], 0x07030602u);
    P[4 * e] = perm(lo23, lo01, 0x05040100u);
    P[4 * e + 1] = perm(lo23, lo01, 0x07060302u);
    P[4 * e + 2] = perm(hi23, hi01, 0x05040100u);
    P[4 * e + 3] = perm(hi23, hi01, 0x07060302u);
  }
  // Row classes from the number of ones per row (= per bit position, over the 16 packed columns).
  // Carry-save adders (sum = a ^ b ^ c, carry = majority: one v_bitop3_b32 each): b0 = bit 0 of the
  // count, b1 = bit 1, hi = the count is 4 or more.
  u32 c2[8];                      // carries of weight 2
  u32 s0, s1, s2, s3, s4;
#define MPC_FA(S, C, a, b, c) { S = xor3(a, b, c); C = maj3(a, b, c); }
  MPC_FA(s0, c2[0], P[0], P[1], P[2])
  MPC_FA(s1, c2[1], P[3], P[4], P[5])
  MPC_FA(s2, c2[2], P[6], P[7], P[8])
  MPC_FA(s3, c2[3], P[9], P[10], P[11])
  MPC_FA(s4, c2[4], P[12], P[13], P[14])
  u32 s5, s6;
  MPC_FA(s5, c2[5], s0, s1, s2)
  MPC_FA(s6, c2[6], s3, s4, P[15])
  const u32 b0 = s5 ^ s6;
  c2[7] = s5 & s6;
  u32 d0, d1, d2, c4a, c4b, c4c;
  MPC_FA(d0, c4a, c2[0], c2[1], c2[2])
  MPC_FA(d1, c4b, c2[3], c2[4], c2[5])
  MPC_FA(d2, c4c, d0, d1, c2[6])
#undef MPC_FA
  const u32 b1 = d2 ^ c2[7];
  const u32 hi = or3(c4a, c4b, c4c) | (d2 & c2[7]);
  // two ones in adjacent columns c, c+1 (c = 0..14): (P[c] & P[c+1]) | (P[c+1] & P[c+2]) is one bitop3
  constexpr unsigned ADJ = (BO_A & BO_B) | (BO_B & BO_C);
  const u32 A = or3(or3(bitop3<ADJ>(P[0], P[1], P[2]), bitop3<ADJ>(P[2], P[3], P[4]), bitop3<ADJ>(P[4], P[5], P[6])),
                    or3(bitop3<ADJ>(P[6], P[7], P[8]), bitop3<ADJ>(P[8], P[9], P[10]), bitop3<ADJ>(P[10], P[11], P[12])),
                    and_or(P[14], P[15], bitop3<ADJ>(P[12], P[13], P[14])));
  const u32 Sf = or3(or3(P[0], P[1], P[2]), or3(P[3], P[4], P[5]), P[6] | P[7]);        // columns 0..7
  const u32 Sb = or3(or3(P[8], P[9], P[10]), or3(P[11], P[12], P[13]), P[14] | P[15]);  // columns 8..15
  const u32 single = bitop3<(BO_A & ~BO_B & ~BO_C) & 0xFFu>(b0, b1, hi);    // exactly one 1: 7 bits
  const u32 cnt2 = bitop3<(~BO_A & BO_B & ~BO_C) & 0xFFu>(b0, b1, hi);     // exactly two
  const u32 two = cnt2 & A;                                                // ... adjacent: 8 bits
  const u32 rest = bitop3<(BO_A | (BO_B & ~BO_C)) & 0xFFu>(and_or(b0, b1, hi), cnt2, A);   // three or more, or two apart
  const u32 half = bitop3<(BO_A & ~(BO_B & BO_C)) & 0xFFu>(rest, Sf, Sb);  // one 8-column half empty: 12 bits
  const u32 full = bitop3<(BO_A & BO_B & BO_C)>(rest, Sf, Sb);             // 17 bits
  Z = ~(Sf | Sb);
  return 7u * __popc(single) + 8u * __popc(two) + 12u * __popc(half) + 17u * __popc(full);
}

// common encoder (FPCModule.cpp:19-85) over all rows of the line: non-zero rows by
// pattern, zero rows as runs in row order (plane-major, then column group)
template <int W>
__device__ __forceinline__ u32 lane_encode(const u32 (&t)[W])
{
  constexpr int NG = W / 4;
  u32 bits = 0;
  u64 ZP = 0;     // byte j of ZP: zero-row mask of column group j (bit 7-p = plane p)
  if constexpr (NG >= 4) {
#pragma unroll
    for (int h = 0; h < NG / 4; h++) {
      u32 Z;
      bits += encode_rows_x4(&t[16 * h], Z);
      ZP |= (u64)Z << (32 * h);
    }
  } else {
#pragma unroll
    for (int j = 0; j < NG; j++) {
      u32 Z;
      bits += encode_rows(&t[4 * j], Z);
      ZP |= (u64)Z << (8 * j);
    }
  }
  // previous row of (p, j): (p, j-1), or (p-1, NG-1) for j = 0; next row: (p, j+1), or (p+1, 0)
  constexpr u64 all = NG == 8 ? ~0ull : ((1ull << (8 * (NG & 7))) - 1ull);
  const u64 prevP = ((ZP << 8) & all) | (ZP >> (8 * (NG - 1) + 1));
  const u64 nextP = (ZP >> 8) | ((ZP & 0x7full) << (8 * (NG - 1) + 1));
  const u64 starts = ZP & ~prevP;
  return bits + 4u * (u32)__popcll(starts) + 3u * (u32)__popcll(starts & nextP);   // 4 bits per run, 7 if longer than one row
}

// Run-length statistics per lane: consecutive lines of a lane with the same
// (cluster, size) key only touch registers.
struct LaneRun {
  u32 key;     // (cluster + 1) << 16 | size
  u32 cnt;
  u32 acc_r, acc_r2;
  bool enc_hot;   // wave-uniform: the previous group of this wave had lines the common encoder compressed
};

__device__ __forceinline__ void lane_run_flush(const LaneRun &rs, const WgStats &st, int K, int bins)
{
  if (rs.cnt) {
    const int k = (int)(rs.key >> 16);
    atomicAdd(&st.hist[k * bins + (int)(rs.key & 0xffffu)], rs.cnt);
    if (rs.acc_r | rs.acc_r2) {
      atomicAdd(&st.sums[k], (u64)rs.acc_r);
      atomicAdd(&st.sums[K + k], (u64)rs.acc_r2);
    }
  }
}

// per-wave constants and run state of the kernel loop
struct LaneEnv {
  WgStats st;
  int K, bins;
  u32 enc_zero, enc_same, enc_unc;
  u32 n_lines, lane;
  u64 first_line;
  uint16_t *sizes_out;
  int8_t *sel_out;
  u32 *defer_q;      // this wave's queue of deferred lines (LDS, defer_cap entries)
  u32 defer_cap;
};

// sum and sum of squares of the bytes of W words (two chains each: the reductions are
// latency-bound on one accumulator)
template <int W>
__device__ __forceinline__ void byte_sums(const u32 (&w)[W], u32 &sum, u32 &sum2)
{
  u32 a0 = 0, a1 = 0, q0 = 0, q1 = 0;
#pragma unroll
  for (int e = 0; e < W; e += 2) {
    a0 = sum_bytes(w[e], a0);
    q0 = sum_sq_bytes(w[e], q0);
    a1 = sum_bytes(w[e + 1], a1);
    q1 = sum_sq_bytes(w[e + 1], q1);
  }
  sum = a0 + a1;
  sum2 = q0 + q1;
}

__device__ __forceinline__ void lane_run_add(LaneRun &rs, u32 key, u32 sum_r, u32 sum_r2, const LaneEnv &E)
{
  if (key != rs.key) {
    lane_run_flush(rs, E.st, E.K, E.bins);
    rs.key = key;
    rs.cnt = 0;
    rs.acc_r = 0;
    rs.acc_r2 = 0;
  }
  rs.cnt++;
  rs.acc_r += sum_r;
  rs.acc_r2 += sum_r2;
}

// one group of 64 lines, held in v[].  Streaming groups (DRAIN = false): lane i evaluates line
// line0 + i; FULL: all 64 lines exist (every group but the last one of a launch), which keeps EXEC
// untouched on the way; lines that only a rarely useful module could win with may be set aside into
// the wave's queue (qn = entries waiting, wave-uniform).  Drain groups (DRAIN = true): this lane
// evaluates queued line `dline` if `dvalid`, with every module it needs.
// loff: this lane of a streaming group evaluates line line0 + loff; alt: see lane_prefilters.
template <int W, bool OUT, bool GEN, bool FULL, bool DRAIN, int... KINDS>
__device__ __forceinline__ void lane_step(const uint4 (&v)[W / 4], u32 line0, u32 loff, u32 dline, bool dvalid,
                                          const MpcVpcParams &P, const LaneEnv &E, LaneRun &rs, u32 &qn, bool &alt)
{
  constexpr int L = 4 * W;
  constexpr int NQ = W / 4;
  constexpr int NPT = sizeof...(KINDS);
  constexpr u32 uncomp = 8u * L;
  const u32 line = DRAIN ? dline : line0 + loff;
  const bool valid = DRAIN ? dvalid : (FULL ? true : (line < E.n_lines));
  const u64 valid_mask = (FULL && !DRAIN) ? ~0ull : __ballot(valid);
  bool deferred = false;
  Lane<W> c;
  c.gen = GEN;
#pragma unroll
  for (int i = 0; i < NQ; i++) {
    c.x[4 * i] = v[i].x; c.x[4 * i + 1] = v[i].y; c.x[4 * i + 2] = v[i].z; c.x[4 * i + 3] = v[i].w;
  }

  // ---- AllZero / AllWordSame (VPC.cpp:332-364) ----
  bool is_rep = false, is_zero = false;
  if (__ballot(valid && c.x[1] == c.x[0])) {   // wave-uniform: otherwise no line of the wave can be either
    u32 diff = 0;
#pragma unroll
    for (int e = 1; e < W; e++) diff |= c.x[e] ^ c.x[0];
    is_rep = diff == 0;                       // every word equals word 0
    is_zero = (diff | c.x[0]) == 0;
  }
  bool need = valid && !(is_zero || (P.has_aws && is_rep));
  u64 need_mask = __ballot(need);

  int chosen = is_zero ? 0 : 1;
  u32 size = is_zero ? E.enc_zero : E.enc_same;
  u32 sum_r = 0, sum_r2 = 0;        // residue statistics of this line (VPC.cpp:417-443)

  // truncated scan table (run-time loop only): per residue word the bits that are scanned
  ctab_t scan_mask = ((NPT == 0 || (GEN && P.plane_mask == ~0u)) && P.trunc_off >= 0) ? lane_tab(P) + P.trunc_off : nullptr;     // (byte-major: plane_mask is all ones)
  c.trunc = GEN ? scan_mask : nullptr;
  if (need_mask) {     // wave-uniform: some line needs the prediction modules
    c.b0 = c.x[0] & 0xffu;
    c.rootb = perm(c.x[0], c.x[0], 0u);
    LaneBest<W> best;
    if constexpr (NPT > 0) {
      u64 defer_mask = 0;
      const bool allow_defer = !DRAIN && E.defer_cap != 0u && qn + (u32)((NPT - 1) * MPC_DEFER_MAX) <= E.defer_cap;
      u32 keep_bits = 0;
      if ((__ballot(lane_row0_min<W, NPT, 0, KINDS...>(c, P) != 0u) & need_mask) != need_mask)
        keep_bits = lane_prefilters<W, NPT, 0, KINDS...>(c, P, need_mask, defer_mask, allow_defer, alt);
      if (!DRAIN && defer_mask) {          // wave-uniform and rare: some lines leave for the queue
        deferred = (defer_mask >> E.lane) & 1ull;
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(defer_mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)defer_mask, 0u));
        if (deferred) E.defer_q[qn + rank] = line;
        qn += (u32)__popcll(defer_mask);
        need = need && !deferred;
      }
      if (keep_bits == 0) lane_last<W, NPT, 0, KINDS...>(c, P, best);
      else lane_seq<W, NPT, 0, KINDS...>(c, P, keep_bits, best, false);
    } else {
      lane_seq_runtime<W>(c, P, best, scan_mask, need_mask);
    }
    // without any prediction module the empty scanned array encodes to 0 bits and the line is
    // reported uncompressed at that size (VPC.cpp:397-407 with an empty maxScanned)
    const bool no_pred = NPT == 0 && P.n_pred == 0;
    const u32 unc_size = (no_pred ? 0u : uncomp) + E.enc_unc;

    // XOR stage of the winner, on bytes: b ^ (b >> 1), or b ^ 0x7f where the MSB is set;
    // column 0 untouched.  The flavour is usually the same for the whole wave.
    u32 t[W];
    const u64 cx_mask = __ballot(best.cx != 0);
    if ((cx_mask & need_mask) == need_mask) {
#pragma unroll
      for (int e = 0; e < W; e++) t[e] = best.r[e] ^ ((best.r[e] >> 1) & (e == 0 ? 0x7f7f7f00u : L7F));
    } else {
      // both flavours, selected per lane with bit masks (no divergence)
      const u32 cx7 = best.cx ? L7F : 0u, ncx = best.cx ? 0u : ~0u;
#pragma unroll
      for (int e = 0; e < W; e++) {
        const u32 a = best.r[e] & H80;
        const u32 f = ((best.r[e] >> 1) & cx7) | ((a - (a >> 7)) & ncx);
        t[e] = best.r[e] ^ (e == 0 ? (f & 0xffffff00u) : f);
      }
    }
    if ((NPT == 0 || GEN) && scan_mask) {      // (wave-uniform) a table cut inside a bit plane: per-word masks
#pragma unroll
      for (int e = 0; e < W; e++) t[e] &= scan_mask[e];
    }
    if (MPC_JIT_PLANES && NPT > 0) {       // each module its own number of bit planes: the winner's
#pragma unroll
      for (int e = 0; e < W; e++) t[e] &= best.pm;
    }
    if (GEN && P.plane_mask != ~0u) {      // (wave-uniform) a table that stops after whole bit planes: the others are not scanned
#pragma unroll
      for (int e = 0; e < W; e++) t[e] &= P.plane_mask;
    }
    // The certificate pays where it closes every line of the group (incompressible data); where the
    // previous group of this wave still had compressible lines it is skipped and the encoder runs
    // straight away (rs.enc_hot, wave-uniform).
    bool open = need;
    const bool byte_major = (NPT == 0 && P.byte_major) || (MPC_JIT_BM && NPT > 0);    // (run-time loop: no certificate for that order)
    if (!rs.enc_hot && !byte_major) open = need && !lane_certified<W>(t);
    if (MPC_JIT_BM && NPT > 0 && !rs.enc_hot) open = need && !lane_certified_bm<W>(t);
    if constexpr (MPC_ABLATE & 16) {                  // timing ablation: the certificate stays, the compressible path is cut off
      if (__ballot(open)) E.st.hist[0] = 1u;          // (keeps the certificate alive; results are wrong)
      open = false;
    }
    u32 enc = uncomp;
    if (__ballot(open)) {
      if (byte_major) enc = lane_encode_bm<W>(t);
      else enc = lane_encode<W>(t);
    }
    const bool keep = open && enc < uncomp && !no_pred;             // VPC.cpp:397-407
    const u64 keep_mask = __ballot(keep);
    rs.enc_hot = keep_mask != 0;

    // residue statistics over all positions: the winner's residues (the root position
    // holds best.root_r, not the raw root), or the raw bytes when the line stays
    // uncompressed; nothing for AllZero / AllWordSame lines (VPC.cpp:412 is not reached)
    if (keep_mask == 0 && need_mask == valid_mask) {
      // every line of the group stays uncompressed (the common case on incompressible data):
      // one wave-uniform key, sums of the raw bytes
      if constexpr (MPC_ABLATE & 4) { sum_r = c.x[3]; sum_r2 = c.x[5]; }
      else byte_sums<W>(c.x, sum_r, sum_r2);
      if (valid) lane_run_add(rs, unc_size, sum_r, sum_r2, E);       // cluster -1: key = size
      if constexpr (OUT) {
        if (valid) {
          const u64 at = E.first_line + line;
          if (E.sizes_out) E.sizes_out[at] = (uint16_t)unc_size;
          if (E.sel_out) E.sel_out[at] = (int8_t)-1;
        }
      }
      return;
    }
    if (keep_mask == need_mask) {
      // every line that needed the modules keeps its encoding (wave-uniform; the usual case on compressible data): the
      // winner's residues as they are, nothing to select
      const u32 w0 = mask_sel(0xffu, best.root_r, best.r[0]);
      u32 a0 = sum_bytes(w0, 0u), q0 = sum_sq_bytes(w0, 0u), a1 = 0, q1 = 0;
#pragma unroll
      for (int e = 1; e < W; e += 2) {
        a1 = sum_bytes(best.r[e], a1);
        q1 = sum_sq_bytes(best.r[e], q1);
        if (e + 1 < W) {
          a0 = sum_bytes(best.r[e + 1], a0);
          q0 = sum_sq_bytes(best.r[e + 1], q0);
        }
      }
      sum_r = a0 + a1;
      sum_r2 = q0 + q1;
    } else {
      u32 w[W];
      const u32 km = keep ? ~0u : 0u;
#pragma unroll
      for (int e = 0; e < W; e++) w[e] = mask_sel(km, best.r[e], c.x[e]);
      w[0] = mask_sel(km & 0xffu, best.root_r, w[0]);
      byte_sums<W>(w, sum_r, sum_r2);
    }
    sum_r = need ? sum_r : 0u;
    sum_r2 = need ? sum_r2 : 0u;
    if (need) {
      chosen = keep ? P.start + best.q : -1;
      size = keep ? enc + best.encb : unc_size;
    }
  }

  // ---- statistics: run-length per lane ----
  // a deferred line is counted when it is drained; here it goes to a spare histogram slot behind
  // the last cluster (index K * bins, never flushed), which keeps EXEC whole
  u32 key = ((u32)(chosen + 1) << 16) | size;
  if (!DRAIN) key = deferred ? ((u32)E.K << 16) : key;
  if (valid) lane_run_add(rs, key, sum_r, sum_r2, E);

  // ---- per-line outputs (parity mode) ----
  if constexpr (OUT) {
    if (valid && !deferred) {
      const u64 at = E.first_line + line;
      if (E.sizes_out) E.sizes_out[at] = (uint16_t)size;
      if (E.sel_out) E.sel_out[at] = (int8_t)chosen;
    }
  }
}

template <int NQ>
__device__ __forceinline__ void lane_fetch(uint4 (&v)[NQ], const uint4 *__restrict__ lines, u32 line, u32 n_lines)
{
  // clamped: past the end it re-reads the last line (never evaluated: `valid` is false there)
  const uint4 *src = lines + (u64)min(line, n_lines - 1u) * NQ;
#pragma unroll
  for (int i = 0; i < NQ; i++) v[i] = src[i];
}

// ---- line ring in LDS ---------------------------------------------------------------------------------------
// The unrolled-sequence kernels do not load their lines into registers.  Every wave owns one or two STAGES of
// 64 * L bytes in LDS; a group of 64 lines travels from global memory straight into a stage with
// global_load_lds_dwordx4 (no destination registers, 1 KiB per instruction, fully coalesced, non-temporal) and is
// read out transposed, one line per lane, right before it is evaluated; a stage is requested again as soon as it
// has been read.  The LDS image of a stage is lane-linear per instruction (instruction j, lane l -> 16-byte unit
// 64 j + l), so the bank swizzle sits on the SOURCE address: position (line, pos) holds piece pos ^ f(line) of the
// line, f(line) = (line >> log2(16 / NQ)) & (NQ - 1), which makes the ds_read_b128 of a line-per-lane read-out
// conflict-free (MI355X_MICROARCH.md, LDS: ds_read_b128 is served in groups of 16 lanes).
//
// Why a ring and not registers (round 3, tools/dev/ringbw.hip): the register double buffer cost 32
// VGPRs and spilled at 4 waves per SIMD; its lane-per-line loads stream at 6.2 TB/s where these reach 7.0-7.15 TB/s
// (all-zero trace 2.9 -> 2.4 ms per 16 GiB); and with two stages (128 consecutive lines resident) a wave can take
// the even and the odd lines of the block as its two groups at no cost in memory traffic (paired groups, below).
// ---- workgroup shape and LDS plan of the unrolled-sequence kernels ----------------------------------------------
// One LARGE workgroup per CU: 16 waves (8 for 128-byte lines, whose group code needs two waves' registers per SIMD
// lane) share ONE histogram, which leaves the LDS for the rings.  Four 256-thread workgroups per CU would hold four
// histograms (58 KiB of the 160) and could not have two stages per wave.  A workgroup owns a contiguous range of
// the trace and its waves draw blocks of it from a counter in LDS, so that they finish within one group of each
// other (without it a large workgroup waits for its slowest wave: +7 % on random data, +12 % on the mixed trace).
#ifndef MPC_LANE_WAVES
#define MPC_LANE_WAVES 16
#endif
#ifndef MPC_LANE_WAVES_128
#define MPC_LANE_WAVES_128 8
#endif
// (32-byte lines: the group code needs under 96 registers, so 5 waves per SIMD would fit as two workgroups of 10 waves
// per CU -- measured 12 % SLOWER on random data and 23 % on the mixed trace than one workgroup of 16)
#ifndef MPC_LANE_WAVES_32
#define MPC_LANE_WAVES_32 16
#endif
__host__ __device__ constexpr int lane_wpb_of(int L) { return L <= 32 ? MPC_LANE_WAVES_32 : (L <= 64 ? MPC_LANE_WAVES : MPC_LANE_WAVES_128); }
template <int W, int NPT>
__host__ __device__ constexpr int lane_wpb() { return NPT == 0 ? 4 : lane_wpb_of(4 * W); }

// LDS layout: [sums][histogram + spare slot + block counter][queues: wpb x cap][pad to 1 KiB][rings: wpb x stages x 64 L]
struct LaneLds {
  u32 stages;      // ring stages per wave: 2 (0: the rings do not fit beside the histogram -- the run-time loop kernel takes the configuration)
  u32 cap;         // queue entries per wave
  size_t bytes;    // dynamic LDS of the launch
};
__host__ __device__ static inline size_t lane_ring_off(size_t stats_and_queues) { return (stats_and_queues + 1023u) & ~(size_t)1023u; }

static LaneLds lane_lds_plan(const MpcVpcParams *P, int wpb, bool wants_queue)
{
  const size_t stats = lane_stats_smem(P->M + 1, P->hist_bins), SB = 64u * (size_t)P->L, lds = 160u << 10;
  // queue entries per wave: what fits beside the histogram; 0 = no room, no line is ever set aside
  const u32 caps[5] = {wants_queue ? 256u : 0u, wants_queue ? 192u : 0u, wants_queue ? 160u : 0u, wants_queue ? 96u : 0u, 0u};
  for (u32 cap : caps) {
    const size_t bytes = lane_ring_off(stats + (size_t)wpb * cap * sizeof(u32)) + (size_t)wpb * 2u * SB;
    if (bytes <= lds) return LaneLds{2, cap, bytes};
  }
  return LaneLds{0, 0, 0};
}

// n_lines < 2^31 per launch (the host splits larger batches)
// 64-byte lines and shorter: 16 waves per CU, i.e. at most 128 VGPRs
#ifndef MPC_LB_WAVES
#define MPC_LB_WAVES 4
#endif
#ifndef MPC_LB_WAVES_RT
#define MPC_LB_WAVES_RT 3   /* the run-time module loop: 3 waves per SIMD (168 VGPRs) measured best of 2 / 3 / 4 */
#endif
#ifndef MPC_LB_WAVES_128
#define MPC_LB_WAVES_128 2
#endif
// ring_cfg: ring stages per wave (bits 0..7) | queue entries per wave << 8 (lane_lds_plan; the run-time loop ignores it)
// GEN: the general-layout twin (vpc_lane_gen_kernel): RootIndex 0..15 per module, scan tables that stop after a whole
// number of bit planes.  The plain kernels keep their code (and their names in the profiles) untouched.
// smem: the launch's LDS (the built-in kernels: the dynamic allocation; a run-time compiled one: a static array of the
// plan's size, so that its launch through the module API needs no function attribute)
template <int W, bool OUT, bool GEN, int... KINDS>
__device__ __forceinline__ void vpc_lane_body(unsigned char *smem, const uint4 *__restrict__ lines, u32 n_lines, u64 first_line,
                                              const MpcVpcParams &P, uint16_t *__restrict__ sizes_out, int8_t *__restrict__ sel_out,
                                              u64 *gstats, u32 ring_cfg)
{
  constexpr int NQ = W / 4;             // 16-byte pieces per line
  constexpr u32 WPB = lane_wpb<W, sizeof...(KINDS)>();       // waves per workgroup
  LaneEnv E;
  E.K = P.M + 1;
  E.bins = P.hist_bins;
  E.st.sums = reinterpret_cast<u64 *>(smem);
  E.st.hist = reinterpret_cast<u32 *>(smem + 16 * ((2 * E.K * 8 + 15) / 16));
  if (threadIdx.x == 0) E.st.hist[E.K * E.bins + 1] = 0;     // the workgroup's block counter
  stats_init(E.st, E.K, E.bins, threadIdx.x, 64 * (int)WPB);   // ends with __syncthreads()
  E.defer_cap = ring_cfg >> 8;
  E.defer_q = reinterpret_cast<u32 *>(smem + lane_stats_smem(E.K, E.bins)) + (threadIdx.x >> 6) * E.defer_cap;
  E.enc_zero = (u32)P.enc_bits[1];
  E.enc_same = 32u + (u32)P.enc_bits[2];
  E.enc_unc = (u32)P.enc_bits[0];
  E.n_lines = n_lines;
  E.lane = threadIdx.x & 63;
  E.first_line = first_line;
  E.sizes_out = sizes_out;
  E.sel_out = sel_out;

  LaneRun rs = {0xffffffffu, 0, 0, 0, false};
  u32 iter = 0;
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u64 *routes = gstats + (2 * E.K + E.K * E.bins);         // (test library only: route counters behind the raw statistics)
  (void)routes;
  bool alt = false;    // wave-uniform: the prefilters saw lines of two kinds alternate (paired groups)
  uint4 va[NQ];
  u32 qn = 0;          // deferred lines waiting in the wave's queue (wave-uniform)
  if constexpr (sizeof...(KINDS) == 0) {
    // run-time module loop: its body holds every predictor form, so the group code exists ONCE here; lines are
    // loaded into registers, the next group's loads are issued before the current group is evaluated
    uint4 vb[NQ];
    const u32 stride = gridDim.x * WPB * 64u;
    u32 line0 = (blockIdx.x * WPB + wave) * 64u;
    if (line0 < n_lines) lane_fetch<NQ>(va, lines, line0 + E.lane, n_lines);
    while (line0 < n_lines) {
      lane_fetch<NQ>(vb, lines, line0 + stride + E.lane, n_lines);
      lane_step<W, OUT, false, false, false>(va, line0, E.lane, 0u, true, P, E, rs, qn, alt);
      line0 += stride;
#pragma unroll
      for (int i = 0; i < NQ; i++) va[i] = vb[i];
      if ((++iter & 255u) == 0) {
        lane_run_flush(rs, E.st, E.K, E.bins);
        rs.cnt = 0;
        rs.acc_r = 0;
        rs.acc_r2 = 0;
      }
    }
  } else {
    // ---- streaming through the wave's line ring ----
    constexpr u32 SB = 64u * 4u * W;                      // bytes of a stage
    constexpr u32 kNone = 0xffffffffu;
    const u32 defer_high = E.defer_cap >= 64u ? E.defer_cap - 64u : 0u;
    const u32 ring_lds = (u32)lane_ring_off(lane_stats_smem(E.K, E.bins) + WPB * E.defer_cap * sizeof(u32)) + wave * (2u * SB);       // two stages per wave
    u32 lane_off[NQ];
#pragma unroll
    for (int j = 0; j < NQ; j++) lane_off[j] = ring_src_off<NQ>(j, E.lane);
    const u32 rd0 = (ring_lds + 16u * NQ * E.lane) | (16u * ring_swz<NQ>(E.lane));      // piece 0 of line `lane` of stage 0 (ring_lds is 1-KiB aligned)
    // The workgroup owns the blocks [b_begin, b_end) -- a block is 128 consecutive lines, two groups, one per stage --
    // and its waves draw them one at a time from a counter in LDS.  What is left of the trace behind the last whole
    // block (fewer than 128 lines) is evaluated from plain loads by one wave of the launch.
    const u32 n_blocks = n_lines >> 7;
    const u32 per_wg = uni((n_blocks + gridDim.x - 1u) / gridDim.x);      // (the division runs on the vector unit)
    const u32 b_begin = blockIdx.x * per_wg;
    const u32 b_end = b_begin + per_wg < n_blocks ? b_begin + per_wg : n_blocks;
    u32 *ticket = E.st.hist + E.K * E.bins + 1;
    auto claim = [&]() -> u32 {
      u32 t = 0;
      if (E.lane == 0) t = atomicAdd(ticket, 1u);
      const u32 blk = b_begin + uni(t);
      return uni((b_begin < n_blocks && blk < b_end) ? blk : kNone);
    };
    // group h (0 / 1) of block blk into stage h
    auto request = [&](u32 blk, u32 h) { ring_request<NQ>(lane_off, lines + (u64)(2u * blk + h) * (64u * NQ), ring_lds + h * SB); };
    u32 cb = claim();               // the block whose groups are (being brought) in the stages
    if (cb != kNone) {
      request(cb, 0u);
      request(cb, 1u);
    }
    bool paired = false;            // wave-uniform: this block is evaluated as even lines / odd lines
    u32 pair_left = 0;
    bool tail_done = false;
    for (;;) {
      while (cb != kNone && qn <= defer_high) {
        cb = uni(cb);
        qn = uni(qn);
        iter = uni(iter);
        const bool pair_now = uni(paired);
        const u32 nb = claim();                          // the wave's next block (the counter's latency hides behind the waits)
        for (u32 h = 0; h < 2u; h = uni(h + 1u)) {
          // ---- the group's lines: wait for its request, read it out one line per lane, refill what is free ----
          // (stages are contiguous and hold 64 lines each, so line bl of the block lies at 16 NQ bl; its piece j at
          // position j ^ f(line): one XOR per piece on the address of piece 0)
          u32 a0;
          if (pair_now) {
            // lane i takes line 2 i + h: both stages are needed before the first group (h = 1: nothing is outstanding)
            ring_wait_vm<0>();
            const u32 bl = 2u * E.lane + h;
            a0 = (ring_lds + 16u * NQ * bl) | (16u * ring_swz<NQ>(bl & 63u));
          } else {
            // requests complete in issue order (stage 0, then stage 1): the wanted one has landed when at most the NQ
            // instructions of the one request issued after it are outstanding
            if (uni(h == 0u || nb != kNone)) ring_wait_vm<NQ>();
            else ring_wait_vm<0>();
            a0 = rd0 + h * SB;
          }
          u32 a[NQ];
#pragma unroll
          for (int j = 0; j < NQ; j++) a[j] = a0 ^ (16u * (u32)j);
          ring_read<NQ>(va, a);
          if constexpr (!(MPC_ABLATE & 32)) {                             // (timing ablation 32: stale lines are evaluated)
            if (nb != kNone) {
              if (!pair_now) {
                request(nb, h);                                           // stage h is free
              } else if (h == 1u) {                                       // both stages are free
                request(nb, 0u);
                request(nb, 1u);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);                              // (the requests go out before the arithmetic, not after it)
          // (one call site: the group code is large)
          lane_step<W, OUT, GEN, true, false, KINDS...>(va, pair_now ? cb * 128u + h : cb * 128u + h * 64u, pair_now ? 2u * E.lane : E.lane,
                                                   0u, true, P, E, rs, qn, alt);
          if ((++iter & 255u) == 0) {   // wave-uniform: keeps the 32-bit accumulators far from overflow
            lane_run_flush(rs, E.st, E.K, E.bins);
            rs.cnt = 0;
            rs.acc_r = 0;
            rs.acc_r2 = 0;
          }
        }
        // how the next block is evaluated
        alt = uni(alt);
        paired = uni(paired);
        pair_left = uni(pair_left);
        if (MPC_TESTING && E.lane == 0) route_add(routes, pair_now ? MPC_RT_VPC_PAIRED_BLOCKS : MPC_RT_VPC_PLAIN_BLOCKS, 1u);
        if (paired) {
          if (--pair_left == 0u) {                     // probe with plain groups
            paired = false;
            if (MPC_TESTING && E.lane == 0) route_add(routes, MPC_RT_VPC_TO_PLAIN, 1u);
          }
        } else if (alt) {
          paired = true;
          pair_left = kPairProbe;
          if (MPC_TESTING && E.lane == 0) route_add(routes, MPC_RT_VPC_TO_PAIRED, 1u);
        }
        alt = false;
        cb = nb;
      }
      // ---- drain: the queued lines, 64 at a time from the top of the queue; and, by one wave of the launch once its
      // blocks are done, the lines behind the launch's last whole block (fewer than 128).  Both are groups of lines
      // addressed per lane and fetched with plain loads: one copy of the group code serves them.
      const bool tail_mine = !tail_done && cb == kNone && blockIdx.x == gridDim.x - 1u && wave == 0u;
      u32 tail_l0 = n_blocks * 128u;
      for (;;) {
        u32 dline;
        bool dvalid;
        if (qn > 0u) {
          const u32 take = qn < 64u ? qn : 64u;
          qn -= take;
          if (MPC_TESTING && E.lane == 0) {
            route_add(routes, MPC_RT_VPC_DRAINS, 1u);
            route_add(routes, MPC_RT_VPC_DEFERRED, take);
          }
          dvalid = E.lane < take;
          dline = dvalid ? E.defer_q[qn + E.lane] : 0u;
        } else if (tail_mine && tail_l0 < n_lines) {
          dline = tail_l0 + E.lane;
          dvalid = dline < n_lines;
          tail_l0 += 64u;
          if (MPC_TESTING && E.lane == 0) route_add(routes, MPC_RT_VPC_TAIL_GROUPS, 1u);
        } else {
          break;
        }
        lane_fetch<NQ>(va, lines, dline, n_lines);
        lane_step<W, OUT, GEN, false, true, KINDS...>(va, 0u, 0u, dline, dvalid, P, E, rs, qn, alt);
      }
      if (tail_mine) tail_done = true;
      if (cb == kNone && (tail_done || blockIdx.x != gridDim.x - 1u || wave != 0u)) break;
    }
  }
  lane_run_flush(rs, E.st, E.K, E.bins);
  stats_flush(E.st, E.K, E.bins, gstats, (int)(E.lane + 64u * wave), 64 * (int)WPB);
}

#define MPC_LANE_BOUNDS(W, NPT)                                                                                     \
  __launch_bounds__((64 * lane_wpb<W, NPT>()),                                                                      \
                    (W <= 16 ? (NPT > 0 ? (W == 8 && MPC_LANE_WAVES_32 == 10 ? 5 : MPC_LB_WAVES) : MPC_LB_WAVES_RT) : MPC_LB_WAVES_128))
template <int W, bool OUT, int... KINDS>
__global__ void MPC_LANE_BOUNDS(W, (sizeof...(KINDS)))
vpc_lane_kernel(const uint4 *__restrict__ lines, u32 n_lines, u64 first_line, MpcVpcParams P,
                uint16_t *__restrict__ sizes_out, int8_t *__restrict__ sel_out, u64 *gstats, u32 ring_cfg)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  vpc_lane_body<W, OUT, false, KINDS...>(smem, lines, n_lines, first_line, P, sizes_out, sel_out, gstats, ring_cfg);
}
template <int W, bool OUT, int... KINDS>
__global__ void MPC_LANE_BOUNDS(W, (sizeof...(KINDS)))
vpc_lane_gen_kernel(const uint4 *__restrict__ lines, u32 n_lines, u64 first_line, MpcVpcParams P,
                    uint16_t *__restrict__ sizes_out, int8_t *__restrict__ sel_out, u64 *gstats, u32 ring_cfg)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  vpc_lane_body<W, OUT, true, KINDS...>(smem, lines, n_lines, first_line, P, sizes_out, sel_out, gstats, ring_cfg);
}

// Everything below is host code: launchers of the built-in instantiations.  A run-time compilation of one more sequence
// (mpc_jit.h: hiprtc, -DMPC_LANE_JIT) takes the device code above only and adds its own two kernels.
#ifndef MPC_LANE_JIT
#define OB MPC_FK_ONEBASE
#define CS MPC_FK_CONSEC
#define DF MPC_FK_DIFF
#define WT MPC_FK_WEIGHT

template <int... KINDS>
bool lane_seq_matches(const MpcVpcParams *P)
{
  constexpr int n = sizeof...(KINDS);
  const int kinds[n > 0 ? n : 1] = {KINDS...};
  if (P->n_pred != n || n == 0 || P->runtime_only) return false;
  for (int q = 0; q < n; q++) {
    // a periodic-table instantiation (LK_PW / LK_PW2) needs its flag; the plain one runs any table
    // that has a byte gather, i.e. all but the two-words-back ones
    const MpcFastModule &f = P->fm[q];
    if (f.kind != lk_base(kinds[q])) return false;
    if (lk_dist(kinds[q]) ? f.prev_word != lk_dist(kinds[q]) : f.prev_word == 2) return false;
  }
  return true;
}

template <int... KINDS>
hipError_t lane_launch(const uint4 *l, u32 n_lines, u64 first_line, const MpcVpcParams *P, uint16_t *d_sizes,
                       int8_t *d_sel, u64 *d_stats, int grid, hipStream_t stream)
{
  // OUT = per-line outputs requested (parity mode); the statistics-only build has no output code
  // histograms of many clusters x bins need more than the 64 KiB of LDS a kernel gets by default
#define MPC_LAUNCH_K(KERNEL, WV, OUTV)                                                                               \
  {                                                                                                                \
    if (smem > (64u << 10))                                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL<WV, OUTV, KINDS...>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                            \
    hipLaunchKernelGGL((KERNEL<WV, OUTV, KINDS...>), dim3(wgrid), dim3(64 * wpb), smem, stream, l, n_lines,         \
                       first_line, *P, d_sizes, d_sel, d_stats, ring_cfg);                                         \
  }
#define MPC_LAUNCH(WV)                                                                                              \
  {                                                                                                                \
    constexpr int wpb = lane_wpb<WV, sizeof...(KINDS)>();                                                          \
    size_t smem;                                                                                                   \
    u32 ring_cfg = 0;                                                                                              \
    int wgrid;                                                                                                     \
    if (sizeof...(KINDS) == 0) {                                                                                   \
      smem = lane_stats_smem(P->M + 1, P->hist_bins);                                                              \
      wgrid = grid;                                                                                                \
    } else {                                                                                                       \
      const LaneLds plan = lane_lds_plan(P, wpb, WV <= 16);                                                        \
      if (plan.stages == 0) return hipErrorInvalidConfiguration;      /* (mpc_vpc_lane_unrolled said no) */        \
      smem = plan.bytes;                                                                                           \
      ring_cfg = plan.stages | (plan.cap << 8);                                                                    \
      /* `grid` counts 256-thread workgroups, 32 per CU when the trace is long: two large workgroups per CU */     \
      /* (four of 512 threads), one resident; never more than one block of lines per wave would fill */           \
      const u64 want = (n_lines / 64u + (u64)wpb - 1u) / (u64)wpb;                                                 \
      const u64 cap = (u64)grid / (wpb == 16 ? 16u : 8u);      /* 16 waves: 2 per CU; 8 or 10 waves: 4 per CU */  \
      wgrid = (int)(want < cap ? want : cap);                                                                      \
      if (wgrid < 1) wgrid = 1;                                                                                    \
    }                                                                                                              \
    const bool out = d_sizes || d_sel;                                                                             \
    bool gen_done = false;                                                                                         \
    if constexpr (sizeof...(KINDS) > 0) {                                                                          \
      if (P->gen_layout) {                                                                                         \
        if (out) MPC_LAUNCH_K(vpc_lane_gen_kernel, WV, true) else MPC_LAUNCH_K(vpc_lane_gen_kernel, WV, false)     \
        gen_done = true;                                                                                           \
      }                                                                                                            \
    }                                                                                                              \
    if (!gen_done) {                                                                                                       \
      if (out) MPC_LAUNCH_K(vpc_lane_kernel, WV, true) else MPC_LAUNCH_K(vpc_lane_kernel, WV, false)               \
    }                                                                                                              \
  }
  switch (P->L) {
#if defined(MPC_LANE_W)    // one line size per translation unit (parallel build, see the end of this file)
  case 4 * MPC_LANE_W: MPC_LAUNCH(MPC_LANE_W); break;
#else
#ifndef MPC_DEV_ONLY64     // development builds (tools/ab.py): 64-byte lines only (or 128 with MPC_DEV_ALSO128), seconds to compile
  case 32: MPC_LAUNCH(8); break;
  case 128: MPC_LAUNCH(32); break;
#else
#if defined(MPC_DEV_ALSO128)
  case 128: MPC_LAUNCH(32); break;
#endif
#if defined(MPC_DEV_ALSO32)
  case 32: MPC_LAUNCH(8); break;
#endif
#endif
  case 64: MPC_LAUNCH(16); break;
#endif
  default: return hipErrorInvalidValue;
  }
#undef MPC_LAUNCH
#undef MPC_LAUNCH_K
  return hipGetLastError();
}

// the module sequences this kernel is instantiated for
#define DFP (MPC_FK_DIFF | LK_PW)
#define WTP (MPC_FK_WEIGHT | LK_PW)
#define DFQ (MPC_FK_DIFF | LK_PW2)
#define WTQ (MPC_FK_WEIGHT | LK_PW2)
#ifdef MPC_DEV_ONLY64
#define MPC_LANE_SEQUENCES(X) X(OB, CS, DFP, WTP)
#else
// (the last row: the five data-type models of configs.mpc_config -- Bool/INT8, INT16, INT32/64, FP32, FP64 -- and the
// 8-byte-element models alone)
#define MPC_LANE_SEQUENCES(X) \
  X(OB, CS, DFP, WTP) X(OB, CS, DFQ, WTQ) X(OB, CS, DF, WT) X(OB, CS) X(OB) X(CS) X(DF) X(WT) X(DFP) X(WTP) \
  X(DF, DF, DFP, CS, WTQ) X(DFQ) X(WTQ)
#endif

// the line sizes this translation unit holds kernels for
static bool lane_has_line_size(int L)
{
#if defined(MPC_LANE_W)
  return L == 4 * MPC_LANE_W;
#elif defined(MPC_DEV_ONLY64) && defined(MPC_DEV_ALSO128)
  return L == 64 || L == 128;
#elif defined(MPC_DEV_ONLY64) && defined(MPC_DEV_ALSO32)
  return L == 64 || L == 32;
#elif defined(MPC_DEV_ONLY64)
  return L == 64;
#else
  return L == 32 || L == 64 || L == 128;
#endif
}

#endif   // !MPC_LANE_JIT
}  // namespace
#ifndef MPC_LANE_JIT

// Build layout.  Compiled as it is, this file holds every line size (development builds).  The product build
// (cal_22-mpc_amd/build.py) compiles it four times in parallel: -DMPC_LANE_W=8 / 16 / 32 give the kernels of one
// line size each and export mpc_launch_vpc_lane_w8 / _w16 / _w32; -DMPC_LANE_W=0 gives the dispatcher below.
#define MPC_CAT2(a, b) a##b
#define MPC_CAT(a, b) MPC_CAT2(a, b)

#if !defined(MPC_LANE_W) || MPC_LANE_W == 0
// 1 when the module sequence has an unrolled instantiation whose rings fit the LDS (else the run-time loop runs it)
extern "C" int mpc_vpc_lane_unrolled(const MpcVpcParams *P)
{
  const int wpb = lane_wpb_of(P->L);
  if (lane_lds_plan(P, wpb, P->L <= 64).stages == 0) return 0;
#define X(...) if (lane_seq_matches<__VA_ARGS__>(P)) return 1;
  MPC_LANE_SEQUENCES(X)
#undef X
  return 0;
}

// LDS plan of an unrolled launch, whether or not the sequence is built in: bytes (0: the rings do not fit beside the
// histogram) and the kernels' ring_cfg argument.  (mpc_jit.h: the size of a run-time compiled kernel's static LDS array.)
// wpb: the waves per workgroup of the plan -- the built-in kernels' (16 / 16 / 8 for 32- / 64- / 128-byte lines) when the rings
// fit beside the histogram with that many waves, else the largest smaller workgroup they fit with (many clusters: the paper
// figure's five models at 128-byte lines miss by 1 KiB with 8 waves): only a kernel compiled at creation can have that size.
extern "C" size_t mpc_vpc_lane_ring_plan(const MpcVpcParams *P, unsigned *ring_cfg, int *wpb)
{
  const int full = lane_wpb_of(P->L);
  const int cand[5] = {full, full * 7 / 8, full * 3 / 4, full * 5 / 8, full / 2};
  for (int w : cand) {
    const LaneLds plan = lane_lds_plan(P, w, P->L <= 64);
    if (plan.stages == 0) continue;
    if (ring_cfg) *ring_cfg = plan.stages | (plan.cap << 8);
    if (wpb) *wpb = w;
    return plan.bytes;
  }
  return 0;
}

// Launch of a run-time compiled sequence (mpc_jit.h): the grid and the kernel arguments of lane_launch, through the
// module API; the kernels carry their LDS as a static array.
extern "C" hipError_t mpc_launch_vpc_lane_jit(hipFunction_t fn_stats, hipFunction_t fn_lines, const void *d_lines, u64 n_lines,
                                              const MpcVpcParams *P, uint16_t *d_sizes, int8_t *d_sel, u64 *d_stats, int grid,
                                              hipStream_t stream)
{
  unsigned ring_cfg = 0;
  int wpb = 0;
  if (mpc_vpc_lane_ring_plan(P, &ring_cfg, &wpb) == 0) return hipErrorInvalidConfiguration;
  const u64 max_lines = 1ull << 30;     // 32-bit line indices inside the kernel
  hipFunction_t fn = (d_sizes || d_sel) ? fn_lines : fn_stats;
  for (u64 done = 0; done < n_lines; done += max_lines) {
    const u64 take = (n_lines - done) < max_lines ? (n_lines - done) : max_lines;
    const uint4 *l = static_cast<const uint4 *>(d_lines) + done * (u64)(P->L / 16);
    u32 n32 = (u32)take;
    u64 first_line = done;
    const u64 want = (take / 64u + (u64)wpb - 1u) / (u64)wpb;
    const u64 cap = (u64)grid / (wpb > 8 ? 16u : 8u);
    int wgrid = (int)(want < cap ? want : cap);
    if (wgrid < 1) wgrid = 1;
    MpcVpcParams params = *P;
    void *args[] = {&l, &n32, &first_line, &params, &d_sizes, &d_sel, &d_stats, &ring_cfg};
    const hipError_t e = hipModuleLaunchKernel(fn, (unsigned)wgrid, 1, 1, 64u * (unsigned)wpb, 1, 1, 0, stream, args, nullptr);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// dynamic LDS of a launch (the host checks it against the 160 KiB of a CU when a handle is created)
extern "C" size_t mpc_vpc_lane_smem(const MpcVpcParams *P)
{
  if (!mpc_vpc_lane_unrolled(P)) return lane_stats_smem(P->M + 1, P->hist_bins);
  const int wpb = lane_wpb_of(P->L);
  return lane_lds_plan(P, wpb, P->L <= 64).bytes;
}
#endif

#if defined(MPC_LANE_W) && MPC_LANE_W == 0
extern "C" hipError_t mpc_launch_vpc_lane_w8(const void *, u64, const MpcVpcParams *, uint16_t *, int8_t *, u64 *, int, hipStream_t);
extern "C" hipError_t mpc_launch_vpc_lane_w16(const void *, u64, const MpcVpcParams *, uint16_t *, int8_t *, u64 *, int, hipStream_t);
extern "C" hipError_t mpc_launch_vpc_lane_w32(const void *, u64, const MpcVpcParams *, uint16_t *, int8_t *, u64 *, int, hipStream_t);

extern "C" hipError_t mpc_launch_vpc_lane(const void *d_lines, u64 n_lines, const MpcVpcParams *P, uint16_t *d_sizes,
                                          int8_t *d_sel, u64 *d_stats, int grid, hipStream_t stream)
{
  switch (P->L) {
  case 32: return mpc_launch_vpc_lane_w8(d_lines, n_lines, P, d_sizes, d_sel, d_stats, grid, stream);
  case 64: return mpc_launch_vpc_lane_w16(d_lines, n_lines, P, d_sizes, d_sel, d_stats, grid, stream);
  case 128: return mpc_launch_vpc_lane_w32(d_lines, n_lines, P, d_sizes, d_sel, d_stats, grid, stream);
  default: return hipErrorInvalidValue;
  }
}
#else
#if defined(MPC_LANE_W)
#define MPC_LAUNCH_NAME MPC_CAT(mpc_launch_vpc_lane_w, MPC_LANE_W)
#else
#define MPC_LAUNCH_NAME mpc_launch_vpc_lane
#endif
extern "C" hipError_t MPC_LAUNCH_NAME(const void *d_lines, u64 n_lines, const MpcVpcParams *P, uint16_t *d_sizes,
                                      int8_t *d_sel, u64 *d_stats, int grid, hipStream_t stream)
{
  if (!lane_has_line_size(P->L)) return hipErrorInvalidValue;
  const u64 max_lines = 1ull << 30;     // 32-bit line indices inside the kernel
  // a sequence whose rings do not fit the LDS beside its histogram runs the run-time module loop
  const int wpb = lane_wpb_of(P->L);
  const bool rings_fit = lane_lds_plan(P, wpb, P->L <= 64).stages != 0;
  for (u64 done = 0; done < n_lines; done += max_lines) {
    const u64 take = (n_lines - done) < max_lines ? (n_lines - done) : max_lines;
    const uint4 *l = static_cast<const uint4 *>(d_lines) + done * (u64)(P->L / 16);
    hipError_t e = hipErrorInvalidValue;
    bool launched = false;
#define X(...)                                                                                                     \
    if (!launched && rings_fit && lane_seq_matches<__VA_ARGS__>(P)) {                                              \
      e = lane_launch<__VA_ARGS__>(l, (u32)take, done, P, d_sizes, d_sel, d_stats, grid, stream);                  \
      launched = true;                                                                                             \
    }
    MPC_LANE_SEQUENCES(X)
#undef X
    if (!launched) e = lane_launch<>(l, (u32)take, done, P, d_sizes, d_sel, d_stats, grid, stream);   // run-time sequence
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
#endif
#endif   // !MPC_LANE_JIT
