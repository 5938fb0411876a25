// mpc_config.h -- VPC configuration model: JSON -> modules -> kernel parameters.
//
// Mirrors what VPC::parseConfig builds (reference src/compressor/VPC.cpp:72-330),
// validates what the reference silently assumes (so that no run depends on
// undefined behaviour), and classifies the configuration onto the fast or the
// generic kernel.  Host-only C++17; no HIP types here.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "mpc_device.h"
#include "mpc_json.h"

namespace mpc {

enum ModuleKind { MOD_ALLZERO = 0, MOD_ALLWORDSAME = 1, MOD_PREDCOMP = 2 };
enum PredKind { PRED_WEIGHT = 0, PRED_DIFF = 1, PRED_ONEBASE = 2, PRED_CONSEC = 3 };

struct Module {
  int kind = MOD_ALLZERO;
  int pred_kind = 0;
  int root = 0;
  bool consecutive_xor = false;
  std::vector<int> base;
  std::vector<float> weight;
  std::vector<int> diff;
  int table_size = 0;
  std::vector<int> rows, cols;
};

struct VpcConfig {
  int M = 0;
  int L = 0;
  std::vector<int> enc_bits;  // index cluster+1, M+1 entries
  std::vector<Module> modules;
  bool has_aws = false;
  int start = 1;   // first PredComp module index
  int n_pred = 0;
  int hist_bins = 0;
};

// (int)log2f(weight), PredictorModule.cpp:30
inline int weight_shift(float w) { return (int)log2f(w); }

inline bool read_file(const std::string &path, std::string &out)
{
  std::ifstream f(path, std::ios::in | std::ios::binary);
  if (!f.is_open()) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}

// Returns 0 or a negative MPC_E_* style code (-74 parse/schema, -22 unsupported value).
inline int parse_vpc_config(const std::string &text, VpcConfig &cfg, std::string &err)
{
  mpcjson::Value root;
  if (!mpcjson::parse(text, root, err)) {
    err = "not a valid json file: " + err;
    return -74;
  }
  bool ok = true;
  const mpcjson::Value &ov = root["overview"];
  cfg.M = ov["num_modules"].asInt(ok);
  cfg.L = ov["lineSize"].asInt(ok);
  if (!ok) { err = "overview.num_modules / overview.lineSize are not integers"; return -74; }
  if (cfg.M < 1 || cfg.M > MPC_MAX_MODULES) { err = "overview.num_modules must be in 1.." + std::to_string(MPC_MAX_MODULES); return -22; }
  if (cfg.L < 4 || cfg.L > MPC_MAX_LINE || (cfg.L % 4)) { err = "overview.lineSize must be a multiple of 4 in 4.." + std::to_string(MPC_MAX_LINE); return -22; }
  const int M = cfg.M, L = cfg.L;
  cfg.enc_bits.assign((size_t)M + 1, 0);
  if (ov["encoding_bits"].isNull()) {
    // (int)ceil(log2f((float)m_NumClusters)), VPC.cpp:104
    int eb = (int)ceil(log2f((float)(M + 1)));
    for (int k = 0; k <= M; k++) cfg.enc_bits[(size_t)k] = eb;
  } else {
    for (int k = 0; k <= M; k++) {
      int v = ov["encoding_bits"][k].asInt(ok);
      if (!ok || v < 0 || v > 4096) { err = "overview.encoding_bits entries must be small non-negative integers"; return -22; }
      cfg.enc_bits[(size_t)k] = v;
    }
  }
  cfg.modules.assign((size_t)M, Module());
  for (int i = 0; i < M; i++) {
    const mpcjson::Value &spec = root["modules"][std::to_string(i)];
    Module &m = cfg.modules[(size_t)i];
    std::string name = spec["name"].asString(ok);
    if (!ok) { err = "modules." + std::to_string(i) + ".name is not a string"; return -74; }
    if (name == "PredComp") {
      m.kind = MOD_PREDCOMP;
      const mpcjson::Value &sub = spec["submodules"];
      const mpcjson::Value &p = sub["ResidueModule"]["PredictorModule"];
      std::string pname = p["name"].asString(ok);
      int pl = p["LineSize"].asInt(ok);
      m.root = p["RootIndex"].asInt(ok);
      if (!ok) { err = "PredictorModule fields of module " + std::to_string(i) + " have the wrong type"; return -74; }
      if (pname == "WeightBasePredictor") m.pred_kind = PRED_WEIGHT;
      else if (pname == "DiffBasePredictor") m.pred_kind = PRED_DIFF;
      else if (pname == "OneBasePredictor") m.pred_kind = PRED_ONEBASE;
      else if (pname == "ConsecutiveBasePredictor") m.pred_kind = PRED_CONSEC;
      else { err = "\"" + pname + "\" is not a valid predictor module. Check the config file."; return -22; }
      if (pl != L) { err = "PredictorModule.LineSize (" + std::to_string(pl) + ") differs from overview.lineSize (" + std::to_string(L) + "): not supported"; return -22; }
      if (m.root < 0 || m.root >= L) { err = "RootIndex out of range in module " + std::to_string(i); return -22; }
      if (m.pred_kind == PRED_CONSEC && m.root != 0) {
        // PredictorModule.cpp:167 reads inputLine[-1] for any other root
        err = "ConsecutiveBasePredictor needs RootIndex 0"; return -22;
      }
      if (m.pred_kind == PRED_WEIGHT || m.pred_kind == PRED_DIFF) {
        m.base.assign((size_t)L, 0);
        m.weight.assign((size_t)L, 1.0f);
        m.diff.assign((size_t)L, 0);
        for (int j = 0; j < L; j++) {
          m.base[(size_t)j] = p["BaseIndexTable"][j].asInt(ok);
          if (m.pred_kind == PRED_WEIGHT) m.weight[(size_t)j] = p["WeightTable"][j].asFloat(ok);
          else m.diff[(size_t)j] = p["DiffTable"][j].asInt(ok);
          if (!ok) { err = "predictor table entry of the wrong type in module " + std::to_string(i); return -74; }
          if (j == m.root) continue;
          if (m.base[(size_t)j] < 0 || m.base[(size_t)j] >= L) { err = "BaseIndexTable entry out of range in module " + std::to_string(i); return -22; }
          if (m.pred_kind == PRED_WEIGHT) {
            float w = m.weight[(size_t)j];
            if (!(w > 0.0f) || std::isinf(w)) { err = "WeightTable entries must be positive and finite (module " + std::to_string(i) + ")"; return -22; }
            int s = weight_shift(w);
            if (s < -31 || s > 31) { err = "WeightTable entry out of range (module " + std::to_string(i) + ")"; return -22; }
          }
        }
      }
      m.consecutive_xor = sub["XORModule"]["consecutiveXOR"].asBool(ok);
      const mpcjson::Value &sc = sub["ScanModule"];
      m.table_size = sc["TableSize"].asInt(ok);
      if (!ok) { err = "XORModule/ScanModule fields of module " + std::to_string(i) + " have the wrong type"; return -74; }
      if (m.table_size < 0 || m.table_size > 8 * L) { err = "ScanModule.TableSize must be within 0..8*lineSize (module " + std::to_string(i) + ")"; return -22; }
      m.rows.assign((size_t)m.table_size, 0);
      m.cols.assign((size_t)m.table_size, 0);
      for (int j = 0; j < m.table_size; j++) {
        int r = sc["Rows"][j].asInt(ok), c = sc["Cols"][j].asInt(ok);
        if (!ok || r < 0 || r >= 8 || c < 0 || c >= L) { err = "ScanModule Rows/Cols entry out of range (module " + std::to_string(i) + ")"; return -22; }
        m.rows[(size_t)j] = r;
        m.cols[(size_t)j] = c;
      }
      // FPCModule: parsed by the reference, validated for names, then unused (VPC.cpp:209-306)
      const mpcjson::Value &fp = sub["FPCModule"];
      int npat = fp["num_modules"].asInt(ok);
      for (int j = 0; j < npat && j < 64; j++) {
        std::string pn = fp[std::to_string(j)]["name"].asString(ok);
        if (pn != "ZerosPattern" && pn != "SingleOnePattern" && pn != "TwoConsecutiveOnesPattern" &&
            pn != "MaskingPattern" && pn != "UncompressedPattern") {
          err = "\"" + pn + "\" is not a valid pattern module. Check the config file.";
          return -22;
        }
      }
    } else if (name == "AllZero") {
      m.kind = MOD_ALLZERO;
    } else if (name == "ByteplaneAllSame" || name == "AllWordSame") {
      m.kind = MOD_ALLWORDSAME;
    } else {
      err = "\"" + name + "\" is not a valid compression module. Check the config file.";
      return -22;
    }
  }
  // layout rule implied by the unchecked static_casts (VPC.cpp:336,353,376)
  if (cfg.modules[0].kind != MOD_ALLZERO) { err = "module 0 must be AllZero"; return -22; }
  cfg.has_aws = (M > 1 && cfg.modules[1].kind == MOD_ALLWORDSAME);
  cfg.start = cfg.has_aws ? 2 : 1;
  for (int i = 1; i < M; i++) {
    int k = cfg.modules[(size_t)i].kind;
    if (k == MOD_ALLZERO) { err = "AllZero is only valid as module 0"; return -22; }
    if (k == MOD_ALLWORDSAME && i != 1) { err = "AllWordSame is only valid as module 1"; return -22; }
  }
  cfg.n_pred = M - cfg.start;
  if (cfg.n_pred > MPC_MAX_PRED) { err = "too many prediction modules"; return -22; }
  int eb = 0;
  for (int v : cfg.enc_bits) eb = v > eb ? v : eb;
  // VPC.h:21-22: histogram pre-seeded 0..287, grows on demand
  cfg.hist_bins = 8 * L + eb + 1;
  if (cfg.hist_bins < 288) cfg.hist_bins = 288;
  return 0;
}

// ---------------------------------------------------------------------------
// kernel plan
// ---------------------------------------------------------------------------

struct VpcPlan {
  bool fast = false;
  std::string why_generic;         // reason the fast path was not taken
  MpcVpcParams params;             // tab / gtab left null; filled by the caller
  std::vector<uint32_t> tab;       // fast-path dword tables
  std::vector<uint8_t> gtab;       // generic-path byte tables
};

// plane-major identity scan table, possibly truncated (TableSize < 8 L: only the leading bits are scanned,
// the remaining cells of the scanned array stay zero, ScanModule.cpp:13-19)
inline bool is_plane_major_scan(const Module &m, int L)
{
  if (m.table_size > 8 * L) return false;
  for (int i = 0; i < m.table_size; i++)
    if (m.rows[(size_t)i] != i / L || m.cols[(size_t)i] != i % L) return false;
  return true;
}

// byte-major scan table, possibly truncated: scanned bit i = plane i % 8 of byte i / 8
inline bool is_byte_major_scan(const Module &m, int L)
{
  if (m.table_size > 8 * L) return false;
  for (int i = 0; i < m.table_size; i++)
    if (m.rows[(size_t)i] != i % 8 || m.cols[(size_t)i] != i / 8) return false;
  return true;
}

inline void build_vpc_plan(const VpcConfig &cfg, VpcPlan &plan)
{
  const int L = cfg.L, W = L / 4;
  MpcVpcParams &P = plan.params;
  std::memset(&P, 0, sizeof(P));
  P.L = L;
  P.M = cfg.M;
  P.n_pred = cfg.n_pred;
  P.start = cfg.start;
  P.has_aws = cfg.has_aws ? 1 : 0;
  P.hist_bins = cfg.hist_bins;
  for (int k = 0; k <= cfg.M; k++) P.enc_bits[k] = cfg.enc_bits[(size_t)k];
  // ---- generic tables (always built: the generic kernel handles anything) ----
  plan.gtab.clear();
  for (int q = 0; q < cfg.n_pred; q++) {
    const Module &m = cfg.modules[(size_t)(cfg.start + q)];
    MpcGenModule &g = P.gm[q];
    g.pred_kind = m.pred_kind;
    g.root = m.root;
    g.cx = m.consecutive_xor ? 1 : 0;
    g.table_size = m.table_size;
    auto align = [&]() { while (plan.gtab.size() % 4) plan.gtab.push_back(0); };
    align();
    g.off_base = (int32_t)plan.gtab.size();
    for (int j = 0; j < L; j++) plan.gtab.push_back(m.base.empty() ? 0 : (uint8_t)m.base[(size_t)j]);
    g.off_shift = (int32_t)plan.gtab.size();
    for (int j = 0; j < L; j++) {
      int s = 0;
      if (m.pred_kind == PRED_WEIGHT && j != m.root) s = weight_shift(m.weight[(size_t)j]);
      if (s > 8) s = 8;    // any shift >= 8 clears a byte
      if (s < -8) s = -8;
      plan.gtab.push_back((uint8_t)(int8_t)s);
    }
    g.off_diff = (int32_t)plan.gtab.size();
    for (int j = 0; j < L; j++) plan.gtab.push_back(m.diff.empty() ? 0 : (uint8_t)m.diff[(size_t)j]);
    g.off_rows = (int32_t)plan.gtab.size();
    for (int j = 0; j < m.table_size; j++) plan.gtab.push_back((uint8_t)m.rows[(size_t)j]);
    g.off_cols = (int32_t)plan.gtab.size();
    for (int j = 0; j < m.table_size; j++) plan.gtab.push_back((uint8_t)m.cols[(size_t)j]);
    while (plan.gtab.size() % 4) plan.gtab.push_back(0);
    g.off_scan = (int32_t)plan.gtab.size();
    for (int j = 0; j < m.table_size; j++) {
      plan.gtab.push_back((uint8_t)m.cols[(size_t)j]);
      plan.gtab.push_back((uint8_t)m.rows[(size_t)j]);
    }
  }
  while (plan.gtab.size() % 16) plan.gtab.push_back(0);
  P.gtab_bytes = (int32_t)plan.gtab.size();

  // ---- fast-path classification ----
  plan.fast = true;
  plan.why_generic.clear();
  plan.tab.clear();
  auto no = [&](const std::string &why) { plan.fast = false; if (plan.why_generic.empty()) plan.why_generic = why; };
  if (!(L == 32 || L == 64 || L == 128)) no("lineSize not in {32,64,128}");
  for (int q = 0; q < cfg.n_pred && plan.fast; q++) {
    const Module &m = cfg.modules[(size_t)(cfg.start + q)];
    MpcFastModule &f = P.fm[q];
    const std::string tag = "module " + std::to_string(cfg.start + q) + ": ";
    // one scan order for the whole configuration: plane-major, or byte-major.  Entry i of a plane-major table is (row 0,
    // column i), of a byte-major one (row i, column 0), so only a table of at most ONE entry is both; the plane-major test
    // runs first, and every module must have module 0's order and TableSize.
    const Module &m0 = cfg.modules[(size_t)cfg.start];
    const bool bm0 = !is_plane_major_scan(m0, L) && is_byte_major_scan(m0, L);
    if (!(bm0 ? is_byte_major_scan(m, L) : is_plane_major_scan(m, L))) {
      no(tag + "scan table is neither a (truncated) plane-major nor a (truncated) byte-major identity, or the modules differ");
      break;
    }
    if (m.table_size != m0.table_size) {
      // different sizes: only when every plane-major table stops after a whole number of bit planes (a per-module mask)
      auto whole = [&](const Module &x) { return x.table_size == 8 * L || (x.table_size >= L && x.table_size % L == 0); };
      if (bm0 || !whole(m) || !whole(m0)) { no(tag + "scan tables of different sizes"); break; }
      P.planes_differ = 1;
    }
    P.byte_major = bm0 ? 1 : 0;
    if (bm0) P.runtime_only = 1;
    // any root for OneBase / DiffBase / WeightBase: the residue array is the natural one with bytes 0..root
    // rotated by one position (ResidueModule.cpp:24-39).  Roots 1..15 (the rotation stays inside the first row of the
    // scanned array) and truncated plane-major tables whose row 0 is complete (TableSize >= 16) run on the unrolled
    // kernels' general-layout twins; roots above 15, the byte-major order and tables of fewer than 16 entries on the
    // run-time module loop
    f.root = m.root;
    if (m.root != 0 || m.table_size != 8 * L) {
      if (m.root <= 15 && !bm0 && m.table_size >= 16) P.gen_layout = 1;
      else P.runtime_only = 1;
    }
    f.cx = m.consecutive_xor ? 1 : 0;
    f.plane_mask = (bm0 || m.table_size >= 8 * L || m.table_size < L) ? ~0u : ((0xff00u >> (m.table_size / L)) & 0xffu) * 0x01010101u;
    f.tab_off = (int32_t)plan.tab.size();
    if (m.pred_kind == PRED_ONEBASE) { f.kind = MPC_FK_ONEBASE; continue; }
    if (m.pred_kind == PRED_CONSEC) { f.kind = MPC_FK_CONSEC; continue; }
    // windowed tables: every base byte lies in the own or the previous dword -- or, for 8-byte
    // elements, every byte from word 2 on is predicted by the same byte two words back
    // (BaseIndexTable[i] = i - 8), which the lane-per-line kernel serves straight from registers
    std::vector<uint32_t> sel((size_t)W, 0), c1((size_t)W, 0), c2((size_t)W, 0);
    int shifts[2] = {0, 0}, nshift = 0;
    bool stride2 = L >= 16 && m.root < 8;       // (words 0 and 1 have table entries of their own)
    for (int i = 8; i < L; i++) stride2 = stride2 && m.base[(size_t)i] == i - 8;
    bool wgen = false;                           // more than two shift distances: assembled from the table as constants (mpc_jit.h)
    bool gather = false;                         // a base byte outside the window: gathered with the table as constants (mpc_jit.h)
    for (int i = 0; i < L && plan.fast; i++) {
      const int w = i / 4, k = i % 4;
      if (i == m.root) { sel[(size_t)w] |= 0x0cu << (8 * k); continue; }  // root: predicted byte forced to 0
      const int b = m.base[(size_t)i];
      int s;
      if (stride2 && w >= 2) {
        s = k;                                     // the same byte of the source word (two words back)
      } else if (w == 0) {
        if (b > 3) gather = true;
        s = 4 + (b & 3);
      } else {
        if (b < 4 * (w - 1) || b > 4 * w + 3) gather = true;
        s = (b - 4 * (w - 1)) & 7;
      }
      if (b < 0 || b >= L) { no(tag + "BaseIndexTable entry out of range"); break; }
      sel[(size_t)w] |= (uint32_t)s << (8 * k);
      if (m.pred_kind == PRED_DIFF) {
        c1[(size_t)w] |= (uint32_t)(uint8_t)m.diff[(size_t)i] << (8 * k);
      } else {
        const int sh = weight_shift(m.weight[(size_t)i]);
        if (sh >= 8 || sh <= -8) continue;  // byte is always 0: both masks stay 0
        int cls = -1;
        for (int c = 0; c < nshift; c++) if (shifts[c] == sh) cls = c;
        if (cls < 0) {
          if (nshift == 2) { wgen = true; continue; }
          cls = nshift;
          shifts[nshift++] = sh;
        }
        const uint32_t bm = sh >= 0 ? ((0xFFu << sh) & 0xFFu) : (0xFFu >> (-sh));
        (cls == 0 ? c1 : c2)[(size_t)w] |= bm << (8 * k);
      }
    }
    if (!plan.fast) break;
    if (m.pred_kind == PRED_DIFF) {
      f.kind = MPC_FK_DIFF;
      // the kernel adds the constant as ((b & 0x7f..) + c1) ^ ((b & 0x80..) ^ c2)
      for (size_t w = 0; w < c1.size(); w++) {
        c2[w] = c1[w] & 0x80808080u;
        c1[w] = c1[w] & 0x7f7f7f7fu;
      }
    } else {
      // put an unshifted class first: the kernel then skips its two shift instructions
      if (nshift == 2 && shifts[0] != 0 && shifts[1] == 0) {
        std::swap(shifts[0], shifts[1]);
        std::swap(c1, c2);
      }
      // MPC_FK_WEIGHT: class 1 unshifted, class 2 a right shift (or absent); anything else is WEIGHT2
      f.kind = (!wgen && shifts[0] == 0 && shifts[1] <= 0) ? MPC_FK_WEIGHT : MPC_FK_WEIGHT2;
      if (wgen) {
        f.wgen = 1;
        P.wshift_unrolled = 1;
      }
      f.ls1 = shifts[0] > 0 ? shifts[0] : 0;
      f.rs1 = shifts[0] < 0 ? -shifts[0] : 0;
      f.ls2 = shifts[1] > 0 ? shifts[1] : 0;
      f.rs2 = shifts[1] < 0 ? -shifts[1] : 0;
    }
    plan.tab.insert(plan.tab.end(), sel.begin(), sel.end());
    plan.tab.insert(plan.tab.end(), c1.begin(), c1.end());
    plan.tab.insert(plan.tab.end(), c2.begin(), c2.end());
    // c1 split into its low-7-bit and MSB parts (lane-per-line kernel, WeightBase residue)
    for (size_t w = 0; w < c1.size(); w++) plan.tab.push_back(c1[w] & 0x7f7f7f7fu);
    for (size_t w = 0; w < c1.size(); w++) plan.tab.push_back(c1[w] & 0x80808080u);
    if (gather) {
      f.gather = 1;
      P.gather_unrolled = 1;
    }
    if (gather || wgen) stride2 = false;
    // periodic tables: words 1.. all use "the same byte of the previous word" with identical masks /
    // constants, so the kernel needs neither the byte gather nor per-word table entries
    f.prev_word = 1;
    for (size_t w = 1; w < sel.size(); w++)
      if (gather || wgen || sel[w] != 0x03020100u || c1[w] != c1[1] || c2[w] != c2[1]) f.prev_word = 0;
    if (stride2) {
      // two-words-back tables must repeat with the element (2 words) from word 2 on: the kernel
      // has no byte gather for them and reads the entries of words 2 and 3 for even / odd words
      f.prev_word = 2;
      for (size_t w = 4; w < sel.size(); w++)
        if (c1[w] != c1[2 + (w & 1)] || c2[w] != c2[2 + (w & 1)]) {
          // constants that do not repeat every 8 bytes: per-word table entries and a byte gather (two words back is outside the window)
          f.prev_word = 0;
          f.gather = 1;
          P.gather_unrolled = 1;
          break;
        }
    }
  }
  if (P.runtime_only) P.gen_layout = 0;
  if (plan.fast && (P.gather_unrolled || P.wshift_unrolled)) {
    if (P.byte_major || P.runtime_only || P.planes_differ) {
      plan.fast = false;
      plan.why_generic = P.gather_unrolled ? "BaseIndexTable is not windowed (own/previous dword), together with another layout the built-in kernels lack"
                                           : "WeightTable uses more than two shift distances, together with another layout the built-in kernels lack";
    } else {
      P.runtime_only = 1;        // no built-in kernel, and not the run-time loop either (its byte gather is the windowed one)
    }
  }
  if (plan.fast && P.planes_differ) {
    bool roots_ok = true;
    for (int q = 0; q < cfg.n_pred; q++) roots_ok = roots_ok && P.fm[q].root <= 15;
    if (!roots_ok) {
      plan.fast = false;
      plan.why_generic = "scan tables of different sizes together with a RootIndex above 15";
    } else {
      P.runtime_only = 1;        // no built-in kernel, and not the run-time loop either (one mask for all modules there)
      P.gen_layout = 1;
    }
  }
  if (plan.fast && !P.byte_major && cfg.n_pred > 0 && cfg.modules[(size_t)cfg.start].table_size >= 16) {
    bool far_root = false;
    for (int q = 0; q < cfg.n_pred; q++) far_root = far_root || P.fm[q].root > 15;
    if (far_root) {           // (the only reason the built-in twins could not take it)
      P.anyroot_unrolled = 1;
      P.gen_layout = 1;
    }
  }
  if (plan.fast && P.byte_major && cfg.n_pred > 0) {
    bool roots0 = true;
    for (int q = 0; q < cfg.n_pred; q++) roots0 = roots0 && P.fm[q].root == 0;
    const int ts = cfg.modules[(size_t)cfg.start].table_size;
    if (roots0 && ts >= 16) {
      P.bm_unrolled = 1;
      P.gen_layout = ts != 8 * L ? 1 : 0;      // (the truncation masks of the general layout)
    }
  }
  P.plane_mask = ~0u;
  if (plan.fast && cfg.n_pred > 0 && !P.byte_major && !P.planes_differ) {
    const int ts = cfg.modules[(size_t)cfg.start].table_size;
    if (ts >= L && ts % L == 0 && ts < 8 * L) P.plane_mask = ((0xff00u >> (ts / L)) & 0xffu) * 0x01010101u;
  }
  P.trunc_off = -1;
  if (plan.fast && !P.planes_differ && cfg.n_pred > 0 && cfg.modules[(size_t)cfg.start].table_size != 8 * L) {
    // scanned bit i = plane i / L (0 = MSB), byte i % L, for i < TableSize: per residue byte the mask of its scanned bits
    const int ts = cfg.modules[(size_t)cfg.start].table_size;
    while (plan.tab.size() % 4) plan.tab.push_back(0);
    P.trunc_off = (int32_t)plan.tab.size();
    for (int w = 0; w < W; w++) {
      uint32_t word = 0;
      for (int k = 0; k < 4; k++) {
        const int col = 4 * w + k;
        uint32_t bm = 0;
        for (int p = 0; p < 8; p++)
          if ((P.byte_major ? col * 8 + p : p * L + col) < ts) bm |= 0x80u >> p;
        word |= bm << (8 * k);
      }
      plan.tab.push_back(word);
    }
  }
  while (plan.tab.size() % 4) plan.tab.push_back(0);
  if (plan.tab.empty()) plan.tab.assign(4, 0);
  P.tab_words = (int32_t)plan.tab.size();
}

}  // namespace mpc
