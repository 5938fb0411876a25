// mpc_vpc_fast.hip -- the hot kernel: VPC multi-prediction evaluation of lines whose
// configuration has the plane-major scan, RootIndex 0 and "windowed" predictor
// tables (see mpc_config.h: build_vpc_plan).  gfx950 / wave64 only.
//
// Mapping: a line of L bytes is held by LPL = L/16 adjacent lanes, 16 bytes (one
// global_load_dwordx4, fully coalesced) per lane.  Lane g of a line owns columns
// 16g..16g+15 of the residue array, i.e. exactly the 16-bit scanned rows
// (plane p, group g), p = 0..7, of the plane-major scan (reference
// ScanModule.cpp:13-19 with Rows[i]=i/L, Cols[i]=i%L).  All byte arithmetic is
// SWAR on 32-bit words (v_perm_b32 / v_alignbyte / v_sad_u8 / v_dot4_u32_u8);
// cross-lane traffic inside a line is DPP.  Statistics are run-length
// accumulated per lane and only reach LDS / global atomics when the
// (cluster, size) key changes.
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
// The kernel is a template over the sequence of predictor forms so that the
// module loop is unrolled and every table lives in registers; configurations
// whose sequence has no instantiation run the same code with a runtime loop.
#include "mpc_kernel_common.h"

#ifndef MPC_ABLATE
#define MPC_ABLATE 0   // development-only timing ablations (tools/ablate.sh); results are WRONG when non-zero
#endif

struct WinTab { u32 sel[4], c1[4], c2[4]; };
// the same tables for (column group 0, word g): what lane g needs for the row-0 prefilter
struct WinTab0 { u32 sel, c1, c2; };

template <int LPL>
struct LineCtx {
  u32 x[4];    // this lane's 16 bytes
  u32 w0;      // word 0 of the line
  u32 b0;      // byte 0 = root (RootIndex 0)
  u32 rootb;   // root replicated into 4 bytes
  u32 xm1;     // previous lane's last word (garbage for g == 0, never selected)
  u32 g;
  u32 cm0;     // 0xffffff00 on the lane that holds column 0 (the root), else 0xffffffff
};

// 4x4 transpose of 32-bit words across the 4 lanes of a quad: out lane g, slot e
// = in lane e, slot g.
__device__ __forceinline__ void quad_transpose(u32 U[4], u32 lane)
{
  const bool o1 = lane & 1, o2 = lane & 2;
  {
    u32 s0 = o1 ? U[0] : U[1], s1 = o1 ? U[2] : U[3];
    u32 r0 = dpp<QP(1, 0, 3, 2)>(s0), r1 = dpp<QP(1, 0, 3, 2)>(s1);
    U[0] = o1 ? r0 : U[0]; U[1] = o1 ? U[1] : r0;
    U[2] = o1 ? r1 : U[2]; U[3] = o1 ? U[3] : r1;
  }
  {
    u32 s0 = o2 ? U[0] : U[2], s1 = o2 ? U[1] : U[3];
    u32 r0 = dpp<QP(2, 3, 0, 1)>(s0), r1 = dpp<QP(2, 3, 0, 1)>(s1);
    U[0] = o2 ? r0 : U[0]; U[2] = o2 ? U[2] : r0;
    U[1] = o2 ? r1 : U[1]; U[3] = o2 ? U[3] : r1;
  }
}

// Residue bytes (root first == natural order for RootIndex 0) of one module for
// this lane's 16 columns; root_r = residue GetMAE/GetMSE see at the root.
template <int LPL, int KIND>
__device__ __forceinline__ void module_residue(const LineCtx<LPL> &c, const MpcFastModule &fm, const WinTab &wt,
                                               u32 *shuf_line, u32 r[4], u32 &root_r)
{
  constexpr int W = 4 * LPL;
  root_r = 0;
  if constexpr (KIND == MPC_FK_ONEBASE) {
    // predicted = line[0] everywhere; position 0 of the residue array is the raw root
#pragma unroll
    for (int e = 0; e < 4; e++) r[e] = bsub(c.x[e], c.rootb);
    r[0] = (r[0] & c.cm0) | (c.b0 & ~c.cm0);     // position 0 of the residue array is the raw root
  } else if constexpr (KIND == MPC_FK_CONSEC) {
    // inp = bytes reordered plane 3,2,1,0 of each word; predicted[i] = inp[i-1].
    // 4x4 byte transpose of this lane's words: T[k] = byte k of the 4 words
    const u32 lo01 = perm(c.x[1], c.x[0], 0x05010400u), hi01 = perm(c.x[1], c.x[0], 0x07030602u);
    const u32 lo23 = perm(c.x[3], c.x[2], 0x05010400u), hi23 = perm(c.x[3], c.x[2], 0x07030602u);
    u32 in[4];
    if constexpr (LPL == 4) {
      // lane g needs byte (3-g) of all 16 words: slot j <- T[3-j], then transpose the quad
      in[3] = perm(lo23, lo01, 0x05040100u);
      in[2] = perm(lo23, lo01, 0x07060302u);
      in[1] = perm(hi23, hi01, 0x05040100u);
      in[0] = perm(hi23, hi01, 0x07060302u);
      quad_transpose(in, c.g);
    } else {
      // general LPL: scatter into the shuffled line in LDS (byte k of word w -> inp[(3-k)*W + w])
      u32 T[4];
      T[0] = perm(lo23, lo01, 0x05040100u);
      T[1] = perm(lo23, lo01, 0x07060302u);
      T[2] = perm(hi23, hi01, 0x05040100u);
      T[3] = perm(hi23, hi01, 0x07060302u);
#pragma unroll
      for (int k = 0; k < 4; k++) shuf_line[((3 - k) * W + 4 * c.g) / 4] = T[k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int e = 0; e < 4; e++) in[e] = shuf_line[4 * c.g + e];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    const u32 carry = Grp<LPL>::prev(in[3]);     // inp[16g-1]: last byte of the previous lane
    const u32 p0 = alignbyte(in[0], carry, 3) & c.cm0;   // root: predicted byte 0 -> residue[0] = raw root
    r[0] = bsub(c.x[0], p0);
    r[1] = bsub(c.x[1], alignbyte(in[1], in[0], 3));
    r[2] = bsub(c.x[2], alignbyte(in[2], in[1], 3));
    r[3] = bsub(c.x[3], alignbyte(in[3], in[2], 3));
    // predicted[root] = inp[0] = byte 3 of word 0 (PredictorModule.cpp:159-164)
    root_r = (c.b0 - (c.w0 >> 24)) & 0xffu;
  } else {
    // the table forces the predicted root byte to 0, so residue[0] = raw root
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const u32 prevw = (e == 0) ? c.xm1 : c.x[e - 1];
      r[e] = bsub(c.x[e], window_predict<KIND>(c.x[e], prevw, wt.sel[e], wt.c1[e], wt.c2[e], fm));
    }
  }
}

// ---- row-0 prefilter (LPL == 4 or 8) ----------------------------------------
// A module other than the last one can only win the selector with at least one
// leading zero row (ties go to the later module, VPC.cpp:389), i.e. only if the
// MSBs of residue bytes 0..15 are all clear.  Lane g of a line evaluates word g of
// column group 0 (y = that word, yprev = word g-1); returns the lane's MSB flags.
struct Pre4 {
  u32 y, yprev;
  u32 t3, t3prev;   // byte 3 of this lane's 4 words / of the previous lane's (ConsecutiveBase)
};

template <int LPL, int KIND>
__device__ __forceinline__ u32 row0_msbs(const LineCtx<LPL> &c, const Pre4 &p, const MpcFastModule &fm, const WinTab0 &w0)
{
  u32 pred;
  if constexpr (KIND == MPC_FK_ONEBASE) {
    pred = c.rootb;
  } else if constexpr (KIND == MPC_FK_CONSEC) {
    pred = alignbyte(p.t3, p.t3prev, 3);   // inp[4g-1 .. 4g+2]
  } else {
    pred = window_predict<KIND>(p.y, p.yprev, w0.sel, w0.c1, w0.c2, fm);
  }
  pred &= c.cm0;                           // byte 0 of the residue array is the raw root
  const u32 msb = bsub(p.y, pred) & H80;
  return (LPL > 4 && c.g >= 4) ? 0u : msb;   // column group 0 is spread over lanes 0..3 of the line
}

// leading zero rows of the scanned array.  The first non-zero row of the XORed
// planes equals that of the raw residue planes (DESIGN.md "Selector on raw
// residues"), so the selector works on r directly.  lut[S] spreads the 8 plane
// bits of a lane to row order (LPL <= 4).
template <int LPL>
__device__ __forceinline__ u32 leading_zero_rows(const u32 r[4], u32 g, const u32 *lut)
{
  const u32 S = fold8(r[0] | r[1] | r[2] | r[3]);
  if constexpr (LPL <= 4) {
    const u32 F = Grp<LPL>::red_or(lut[S] >> g);   // bit 31-row set <=> row non-zero
    return (u32)__clz((int)F);                      // 32 when every row is zero ...
  } else {
    const u32 O = Grp<LPL>::red_or(S);
    const u32 p_star = (u32)__clz((int)O) - 24u;          // first non-zero plane (0 = MSB); 8 if O == 0
    const u32 has = (S << p_star) & 0x80u;
    const u32 g_star = Grp<LPL>::red_min(has ? g : (u32)LPL);
    return O ? LPL * p_star + g_star : 8u * LPL;
  }
}

__device__ __forceinline__ void load_wintab(const u32 *tab, const MpcFastModule &fm, int W, u32 g, WinTab &wt)
{
  const u32 *t = tab + fm.tab_off;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    wt.sel[e] = t[4 * g + e];
    wt.c1[e] = t[W + 4 * g + e];
    wt.c2[e] = t[2 * W + 4 * g + e];
  }
}

// per-kernel constant state of the compile-time module sequence
template <int NPT>
struct SeqTabs {
  WinTab wt[NPT > 0 ? NPT : 1];
  WinTab0 w0[NPT > 0 ? NPT : 1];
};

struct Winner {
  u32 r[4];
  u32 z;       // run-time sequence only
  int q;
  u32 root_r;
  u32 cx;
};

// ---- compile-time sequence: modules in order; a module that passes the wave-uniform
// row-0 prefilter is skipped (it cannot win); the winner is updated with selects ----
template <int LPL>
__device__ __forceinline__ void take_if_better(Winner &best, const u32 r[4], u32 z, int q, u32 root_r, u32 cx)
{
  const bool take = best.z <= z;     // ties go to the later module (VPC.cpp:389)
  best.z = take ? z : best.z;
  best.q = take ? q : best.q;
  best.root_r = take ? root_r : best.root_r;
  best.cx = take ? cx : best.cx;
#pragma unroll
  for (int e = 0; e < 4; e++) best.r[e] = take ? r[e] : best.r[e];
}

template <int LPL, int NPT, int Q>
__device__ __forceinline__ void eval_seq(const LineCtx<LPL> &, const MpcVpcParams &, const SeqTabs<NPT> &, u32 *,
                                         const u32 *, const Pre4 &, u64, Winner &, bool) {}

template <int LPL, int NPT, int Q, int KIND, int... REST>
__device__ __forceinline__ void eval_seq(const LineCtx<LPL> &c, const MpcVpcParams &P, const SeqTabs<NPT> &tabs,
                                         u32 *shuf_line, const u32 *lut, const Pre4 &pre, u64 need_lines,
                                         Winner &best, bool any_full)
{
  if constexpr (Q + 1 < NPT) {
    bool full = true;
    if constexpr (MPC_ABLATE & 8) full = false;
    if constexpr (LPL >= 4 && !(MPC_ABLATE & 8)) {
      // prefilter: skip the module when no line of the wave can have a leading zero row
      const u32 msb = row0_msbs<LPL, KIND>(c, pre, P.fm[Q], tabs.w0[Q]);
      u64 b = __ballot(msb != 0);
      b |= b >> 1;
      b |= b >> 2;                                   // lead lane of a line: some lane 0..3 has an MSB set
      full = (b & need_lines) != need_lines;         // wave-uniform
      if constexpr (MPC_ABLATE & 8) full = false;
    }
    if (full) {
      u32 r[4], root_r;
      module_residue<LPL, KIND>(c, P.fm[Q], tabs.wt[Q], shuf_line, r, root_r);
      bool contender = true;
      if constexpr (LPL == 2 && !(MPC_ABLATE & 8)) {
        // 32-byte lines: column group 0 is the line's first lane; same row-0 test on the full
        // residue (saves the fold / leading-zero / update work of a module that cannot win)
        const u32 msb = (c.g == 0) ? ((r[0] | r[1] | r[2] | r[3]) & H80) : 0u;
        const u64 b = __ballot(msb != 0);
        contender = (b & need_lines) != need_lines;
      }
      if (contender) {
        const u32 z = leading_zero_rows<LPL>(r, c.g, lut);
        if (any_full) {
          take_if_better<LPL>(best, r, z, Q, root_r, (u32)P.fm[Q].cx);
        } else {   // first candidate of this wave: plain assignment (best is not initialised before)
          best.z = z;
          best.q = Q;
          best.root_r = root_r;
          best.cx = (u32)P.fm[Q].cx;
#pragma unroll
          for (int e = 0; e < 4; e++) best.r[e] = r[e];
        }
        any_full = true;
      }
    }
    eval_seq<LPL, NPT, Q + 1, REST...>(c, P, tabs, shuf_line, lut, pre, need_lines, best, any_full);
  } else {
    // the last module is always evaluated; its z only matters against an earlier candidate
    u32 r[4], root_r;
    if constexpr (MPC_ABLATE & 4) { r[0] = c.x[0]; r[1] = c.x[1]; r[2] = c.x[2]; r[3] = c.x[3]; root_r = 0; }
    else module_residue<LPL, KIND>(c, P.fm[Q], tabs.wt[Q], shuf_line, r, root_r);
    if (any_full) {
      take_if_better<LPL>(best, r, leading_zero_rows<LPL>(r, c.g, lut), Q, root_r, (u32)P.fm[Q].cx);
    } else {
      best.q = Q;
      best.root_r = root_r;
      best.cx = (u32)P.fm[Q].cx;
#pragma unroll
      for (int e = 0; e < 4; e++) best.r[e] = r[e];
    }
  }
}

template <int LPL, int KIND>
__device__ __forceinline__ void eval_update(const LineCtx<LPL> &c, const MpcFastModule &fm, const WinTab &wt,
                                            u32 *shuf_line, const u32 *lut, int q, Winner &best)
{
  u32 r[4], root_r;
  module_residue<LPL, KIND>(c, fm, wt, shuf_line, r, root_r);
  take_if_better<LPL>(best, r, leading_zero_rows<LPL>(r, c.g, lut), q, root_r, (u32)fm.cx);
}

// Run-length statistics.  Every lane of a line sees the same (cluster, size) key; all
// lanes accumulate the byte sums of their own 16 columns, the lead lane (g == 0) counts lines.
struct RunStats {
  u32 key;     // (cluster + 1) << 16 | size
  u32 cnt;
  u32 acc_r, acc_r2;
};

__device__ __forceinline__ void run_flush(const RunStats &rs, const WgStats &st, int K, int bins, bool lead)
{
  if (rs.cnt) {
    const int k = (int)(rs.key >> 16);
    if (lead) atomicAdd(&st.hist[k * bins + (int)(rs.key & 0xffffu)], rs.cnt);
    if (rs.acc_r | rs.acc_r2) {
      atomicAdd(&st.sums[k], (u64)rs.acc_r);
      atomicAdd(&st.sums[K + k], (u64)rs.acc_r2);
    }
  }
}

// any lane of a line set -> every lane of that line set (scalar-unit work on a ballot)
template <int LPL>
__device__ __forceinline__ u64 line_any(u64 b)
{
  constexpr u64 lead = LPL == 2 ? 0x5555555555555555ull : (LPL == 4 ? 0x1111111111111111ull : 0x0101010101010101ull);
  b |= b >> 1;
  if (LPL >= 4) b |= b >> 2;
  if (LPL >= 8) b |= b >> 4;
  b &= lead;
  b |= b << 1;
  if (LPL >= 4) b |= b << 2;
  if (LPL >= 8) b |= b << 4;
  return b;
}

// n_chunks = n_lines * LPL must be below 2^31 (the host splits larger batches)
template <int LPL, bool OUT, int... KINDS>
__global__ void __launch_bounds__(256)
vpc_fast_kernel(const uint4 *__restrict__ lines, u32 n_chunks, u64 first_line, MpcVpcParams P,
                uint16_t *__restrict__ sizes_out, int8_t *__restrict__ sel_out, u64 *gstats)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int L = 16 * LPL;
  constexpr int W = L / 4;              // words per line
  constexpr int NPT = sizeof...(KINDS); // 0: module sequence known only at run time
  constexpr int kinds[NPT > 0 ? NPT : 1] = {KINDS...};
  const int K = P.M + 1, bins = P.hist_bins;
  WgStats st;
  st.sums = reinterpret_cast<u64 *>(smem);
  st.hist = reinterpret_cast<u32 *>(smem + 16 * ((2 * K * 8 + 15) / 16));
  // row-order spread table for leading_zero_rows, then per-wave scratch for the
  // byte-plane shuffle when LPL != 4 (64 lanes x 16 B)
  u32 *lut = reinterpret_cast<u32 *>(smem + vpc_stats_smem(K, bins));
  u32 *shuf = lut + 256 + (threadIdx.x >> 6) * 256;
  {
    // bit j of S (plane 7-j) -> row LPL*(7-j) of lane 0, i.e. bit 31 - LPL*(7-j)
    u32 v = 0;
    for (int j = 0; j < 8; j++)
      if (threadIdx.x & (1u << j)) v |= 1u << ((31 - LPL * (7 - j)) & 31);
    lut[threadIdx.x & 255] = v;
  }
  stats_init(st, K, bins);   // ends with __syncthreads()

  const u32 lane = threadIdx.x & 63;
  const u32 g = lane & (LPL - 1);      // column group inside the line
  u32 *shuf_line = shuf + (lane / LPL) * W;
  const u32 colmask0 = (g == 0) ? 0xffffff00u : 0xffffffffu;   // column 0 is exempt from the XOR stage
  const u32 uncomp = 8u * L;
  const int n_pred = NPT > 0 ? NPT : P.n_pred;
  const u32 enc_zero = (u32)P.enc_bits[1], enc_same = 32u + (u32)P.enc_bits[2], enc_unc = (u32)P.enc_bits[0];
  const u32 cert_min = (u32)P.cert_min17;

  // loop-invariant predictor tables in registers (compile-time sequence only)
  SeqTabs<NPT> tabs;
  if constexpr (NPT > 0) {
#pragma unroll
    for (int q = 0; q < NPT; q++) {
      if (kinds[q] == MPC_FK_DIFF || kinds[q] == MPC_FK_WEIGHT || kinds[q] == MPC_FK_WEIGHT2) {
        load_wintab(P.tab, P.fm[q], W, g, tabs.wt[q]);
        const u32 *t = P.tab + P.fm[q].tab_off;     // column group 0, word g (row-0 prefilter)
        tabs.w0[q].sel = t[g & 3];
        tabs.w0[q].c1 = t[W + (g & 3)];
        tabs.w0[q].c2 = t[2 * W + (g & 3)];
      }
    }
  }

  RunStats rs = {0xffffffffu, 0, 0, 0};
  u32 iter = 0;

  // wave-uniform chunk index: the address math stays on the scalar unit
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 stride = gridDim.x * 256u;
  u32 chunk0 = (blockIdx.x * 4u + wave) * 64u;
  uint4 vnext = make_uint4(0, 0, 0, 0);
  if (chunk0 < n_chunks) vnext = lines[min(chunk0 + lane, n_chunks - 1u)];

  for (; chunk0 < n_chunks; chunk0 += stride) {
    // chunk0 is wave-uniform by construction; readfirstlane makes the compiler keep the
    // whole mask algebra below on the scalar unit
    const u32 rem = n_chunks - __builtin_amdgcn_readfirstlane(chunk0);
    const u64 valid_mask = rem >= 64u ? ~0ull : ((1ull << rem) - 1ull);   // whole line groups
    const uint4 v = vnext;
    // prefetch the next iteration (clamped: past the end it re-reads the last chunk)
    vnext = lines[min(chunk0 + stride + lane, n_chunks - 1u)];

    LineCtx<LPL> c;
    c.x[0] = v.x; c.x[1] = v.y; c.x[2] = v.z; c.x[3] = v.w;
    c.g = g;
    c.cm0 = colmask0;

    // ---- AllZero / AllWordSame: per-line predicates as lane masks, on the scalar unit ----
    c.w0 = Grp<LPL>::first(c.x[0]);
    // a line is all-zero iff every word equals word 0 and word 0 is zero: one line-OR, not two (A/B: -2 %)
    u32 dword_diff = (c.x[0] ^ c.w0) | (c.x[1] ^ c.w0) | (c.x[2] ^ c.w0) | (c.x[3] ^ c.w0);
    asm volatile("" : "+v"(dword_diff));   // keep the OR form: one compare instead of four (A/B: -1.5 %)
    const u64 df_b = __ballot(dword_diff != 0);
    const u64 rep_m = ~line_any<LPL>(df_b);                 // every word of the line equals word 0
    const u64 zero_m = rep_m & __ballot(c.w0 == 0);
    const u64 same_m = P.has_aws ? (rep_m & ~zero_m) : 0ull;
    const u64 need_mask = valid_mask & ~zero_m & ~same_m;
    const bool is_zero = __builtin_amdgcn_inverse_ballot_w64(zero_m);
    const bool need = __builtin_amdgcn_inverse_ballot_w64(need_mask);

    int chosen = is_zero ? 0 : 1;
    u32 size = is_zero ? enc_zero : enc_same;
    u32 rr[4] = {0, 0, 0, 0};          // bytes whose sum / sum of squares go to the residue statistics

    // Skip the prediction modules only when no line of the wave needs them
    // (wave-uniform branch; DPP needs every lane of a line group active).
    if (need_mask) {
      c.b0 = c.w0 & 0xffu;
      c.rootb = perm(c.w0, c.w0, 0u);            // byte 0 replicated
      c.xm1 = Grp<LPL>::prev(c.x[3]);
      Winner best;
      if constexpr (NPT == 0) {
        best.r[0] = best.r[1] = best.r[2] = best.r[3] = 0;
        best.z = 0; best.q = -1; best.root_r = 0; best.cx = 0;
      }
      if constexpr (NPT > 0) {
        Pre4 pre = {0, 0, 0, 0};
        if constexpr (LPL >= 4 && NPT > 1) {
          // word (g & 3) of column group 0 (held by the line's first lane), and its predecessor
          const u32 a0 = Grp<LPL>::first(c.x[0]), a1 = Grp<LPL>::first(c.x[1]);
          const u32 a2 = Grp<LPL>::first(c.x[2]), a3 = Grp<LPL>::first(c.x[3]);
          pre.y = (g & 2) ? ((g & 1) ? a3 : a2) : ((g & 1) ? a1 : a0);
          pre.yprev = (g & 2) ? ((g & 1) ? a2 : a1) : a0;
          bool any_consec = false;
#pragma unroll
          for (int q = 0; q + 1 < NPT; q++) any_consec = any_consec || (kinds[q] == MPC_FK_CONSEC);
          if (any_consec) {
            // inp[0..15] = byte 3 of words 0..15 = lanes 0..3 of the line (W >= 16)
            const u32 hi01 = perm(c.x[1], c.x[0], 0x07030602u), hi23 = perm(c.x[3], c.x[2], 0x07030602u);
            pre.t3 = perm(hi23, hi01, 0x07060302u);
            pre.t3prev = Grp<LPL>::prev(pre.t3);
          }
        }
        constexpr u64 lead_lanes = LPL == 2 ? 0x5555555555555555ull : (LPL == 4 ? 0x1111111111111111ull : 0x0101010101010101ull);
        eval_seq<LPL, NPT, 0, KINDS...>(c, P, tabs, shuf_line, lut, pre, need_mask & lead_lanes, best, false);
      } else {
        for (int q = 0; q < n_pred; q++) {
          const MpcFastModule fm = P.fm[q];
          WinTab w;
          switch (fm.kind) {
          case MPC_FK_ONEBASE: eval_update<LPL, MPC_FK_ONEBASE>(c, fm, w, shuf_line, lut, q, best); break;
          case MPC_FK_CONSEC: eval_update<LPL, MPC_FK_CONSEC>(c, fm, w, shuf_line, lut, q, best); break;
          case MPC_FK_DIFF: load_wintab(P.tab, fm, W, g, w); eval_update<LPL, MPC_FK_DIFF>(c, fm, w, shuf_line, lut, q, best); break;
          case MPC_FK_WEIGHT: load_wintab(P.tab, fm, W, g, w); eval_update<LPL, MPC_FK_WEIGHT>(c, fm, w, shuf_line, lut, q, best); break;
          default: load_wintab(P.tab, fm, W, g, w); eval_update<LPL, MPC_FK_WEIGHT2>(c, fm, w, shuf_line, lut, q, best); break;
          }
        }
      }

      u32 enc = uncomp;
      if ((n_pred > 0) && !(MPC_ABLATE & 2)) {
        // XOR stage of the winner, on bytes: b ^ (b >> 1), or b ^ 0x7f where the MSB is set;
        // column 0 untouched.  The flavour is usually the same for the whole wave.
        u32 t[4];
        const u64 cx_mask = __ballot(best.cx != 0);
        if ((cx_mask & need_mask) == need_mask) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const u32 f = (best.r[e] >> 1) & L7F;
            t[e] = best.r[e] ^ (e == 0 ? (f & colmask0) : f);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const u32 m = (best.r[e] >> 7) & 0x01010101u;
            const u32 f = best.cx ? ((best.r[e] >> 1) & L7F) : ((m << 7) - m);
            t[e] = best.r[e] ^ (e == 0 ? (f & colmask0) : f);
          }
        }
        // Incompressibility certificate: a row whose two 8-column halves are both non-zero
        // and that has a bit outside columns 7/8 costs 17 bits (it is neither a single one
        // nor an adjacent pair).  With cert_min17 such rows the line cannot beat 8*L bits.
        const u32 Sf = fold8(t[0] | t[1]), Sb = fold8(t[2] | t[3]);
        const u32 So = fold8(t[0] | (t[1] & 0x00ffffffu) | (t[2] & 0xffffff00u) | t[3]);
        const u32 n17 = Grp<LPL>::red_add((u32)__popc(Sf & Sb & So));
        const u64 open_mask = __ballot(n17 < cert_min) & need_mask;
        if (open_mask) {
          u32 Z;
          u32 bits = encode_rows(t, Z);
          // zero-row runs in row order r = plane * LPL + g: 4 bits per run, 7 if longer than one row
          const u32 Zprev_lane = Grp<LPL>::prev(Z), Zlast = Grp<LPL>::last(Z);
          const u32 Znext_lane = Grp<LPL>::next(Z), Zfirst = Grp<LPL>::first(Z);
          const u32 prevZ = (g == 0) ? (Zlast >> 1) : Zprev_lane;
          const u32 nextZ = (g == LPL - 1) ? ((Zfirst << 1) & 0xffu) : Znext_lane;
          const u32 starts = Z & ~prevZ;
          bits += 4u * __popc(starts) + 3u * __popc(starts & nextZ);
          enc = Grp<LPL>::red_add(bits);
        }
      }
      // residue statistics over all positions: the winner's residues (the root position
      // holds best.root_r, not the raw root), or the raw bytes when the line stays
      // uncompressed; nothing for AllZero / AllWordSame lines (VPC.cpp:412 is not reached)
      const u64 keep_mask = (n_pred > 0) ? (__ballot(enc < uncomp) & need_mask) : 0ull;
      if (keep_mask == 0 && need_mask == valid_mask) {
        // whole wave uncompressed (the common case on incompressible data)
#pragma unroll
        for (int e = 0; e < 4; e++) rr[e] = c.x[e];
        chosen = -1;
        size = (n_pred > 0 ? uncomp : 0u) + enc_unc;   // no prediction module: the empty array encodes to 0 bits
      } else {
        const bool keep = __builtin_amdgcn_inverse_ballot_w64(keep_mask);
#pragma unroll
        for (int e = 0; e < 4; e++) rr[e] = keep ? best.r[e] : (need ? c.x[e] : 0u);
        if (keep && g == 0) rr[0] = (rr[0] & 0xffffff00u) | best.root_r;
        if (need) {
          chosen = keep ? P.start + best.q : -1;
          size = keep ? enc + (u32)P.enc_bits[chosen + 1] : (n_pred > 0 ? uncomp : 0u) + enc_unc;
        }
      }
    }

    // ---- statistics: run-length per lane ----
    const u32 key = ((u32)(chosen + 1) << 16) | size;
    const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_mask);
    if (valid) {
      if (key != rs.key) {
        run_flush(rs, st, K, bins, g == 0);
        rs.key = key;
        rs.cnt = 0;
        rs.acc_r = 0;
        rs.acc_r2 = 0;
      }
      rs.cnt++;
      if constexpr (!(MPC_ABLATE & 1)) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          rs.acc_r = sum_bytes(rr[e], rs.acc_r);
          rs.acc_r2 = sum_sq_bytes(rr[e], rs.acc_r2);
        }
      }
    }
    if ((++iter & 127u) == 0) {   // wave-uniform: keeps the 32-bit accumulators far from overflow
      run_flush(rs, st, K, bins, g == 0);
      rs.cnt = 0;
      rs.acc_r = 0;
      rs.acc_r2 = 0;
    }

    // ---- per-line outputs (parity mode): one lane per line ----
    if constexpr (OUT) {
      if (valid && g == 0) {
        const u64 line = first_line + (chunk0 + lane) / LPL;
        if (sizes_out) sizes_out[line] = (uint16_t)size;
        if (sel_out) sel_out[line] = (int8_t)chosen;
      }
    }
  }
  run_flush(rs, st, K, bins, g == 0);
  stats_flush(st, K, bins, gstats);
}

// ---------------------------------------------------------------------------
// launcher: pick the instantiation that matches the configuration's sequence
// ---------------------------------------------------------------------------
#define OB MPC_FK_ONEBASE
#define CS MPC_FK_CONSEC
#define DF MPC_FK_DIFF
#define WT MPC_FK_WEIGHT

template <int... KINDS>
static bool seq_matches(const MpcVpcParams *P)
{
  constexpr int n = sizeof...(KINDS);
  const int kinds[n > 0 ? n : 1] = {KINDS...};
  if (P->n_pred != n) return false;
  for (int q = 0; q < n; q++)
    if (P->fm[q].kind != kinds[q]) return false;
  return true;
}

template <int... KINDS>
static hipError_t launch_seq(const uint4 *l, u32 n_chunks, u64 first_line, const MpcVpcParams *P, uint16_t *d_sizes,
                             int8_t *d_sel, u64 *d_stats, int grid, size_t smem, hipStream_t stream)
{
  // persistent grid-stride kernel: `grid` (8 workgroups per CU, capped by the work) was A/B-tested
  // against occupancy-sized grids on one box: no measurable difference, so the simple rule stays
  // OUT = per-line outputs requested (parity mode); the statistics-only build has no output code
#define MPC_LAUNCH(LPLV)                                                                                              \
  if (d_sizes || d_sel)                                                                                              \
    hipLaunchKernelGGL((vpc_fast_kernel<LPLV, true, KINDS...>), dim3(grid), dim3(256), smem, stream, l, n_chunks,    \
                       first_line, *P, d_sizes, d_sel, d_stats);                                                     \
  else                                                                                                               \
    hipLaunchKernelGGL((vpc_fast_kernel<LPLV, false, KINDS...>), dim3(grid), dim3(256), smem, stream, l, n_chunks,   \
                       first_line, *P, d_sizes, d_sel, d_stats)
  switch (P->L) {
  case 32: MPC_LAUNCH(2); break;
  case 64: MPC_LAUNCH(4); break;
  case 128: MPC_LAUNCH(8); break;
  default: return hipErrorInvalidValue;
  }
#undef MPC_LAUNCH
  return hipGetLastError();
}

// statistics + spread table (1 KiB) + shuffle scratch (4 waves x 1 KiB)
extern "C" size_t mpc_vpc_fast_smem(const MpcVpcParams *P) { return vpc_stats_smem(P->M + 1, P->hist_bins) + 5 * 1024; }

extern "C" hipError_t mpc_launch_vpc_fast(const void *d_lines, u64 n_lines, const MpcVpcParams *P, uint16_t *d_sizes,
                                          int8_t *d_sel, u64 *d_stats, int grid, hipStream_t stream)
{
  const size_t smem = mpc_vpc_fast_smem(P);
  const u32 LPL = (u32)P->L / 16u;
  // 32-bit chunk indices inside the kernel: at most 2^30 chunks (16 GiB) per launch
  const u64 max_lines = (1ull << 30) / LPL;
  for (u64 done = 0; done < n_lines; done += max_lines) {
    const u64 take = (n_lines - done) < max_lines ? (n_lines - done) : max_lines;
    const uint4 *l = static_cast<const uint4 *>(d_lines) + done * LPL;
    const u32 n_chunks = (u32)(take * LPL);
    hipError_t e;
#define TRY_SEQ(...)                                                                                   \
    if (seq_matches<__VA_ARGS__>(P))                                                                   \
      e = launch_seq<__VA_ARGS__>(l, n_chunks, done, P, d_sizes, d_sel, d_stats, grid, smem, stream);  \
    else
    TRY_SEQ(OB, CS, DF, WT)   // the 4-predictor "probe" sequence
    TRY_SEQ(OB, CS)
    TRY_SEQ(OB)
    TRY_SEQ(CS)
    TRY_SEQ(DF)
    TRY_SEQ(WT)
      e = launch_seq<>(l, n_chunks, done, P, d_sizes, d_sel, d_stats, grid, smem, stream);   // run-time sequence
#undef TRY_SEQ
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
