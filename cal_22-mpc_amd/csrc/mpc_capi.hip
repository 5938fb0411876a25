// mpc_capi.hip -- the C ABI of include/mpc_hip.h: handles, the pinned
// double-buffered host->device stager, .npy streaming, statistics.
//
// There is no CPU evaluation path in this library: every size it reports was
// computed by a gfx950 kernel in mpc_kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/mpc_hip.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "mpc_config.h"
#include "mpc_device.h"

// libmpc_hip_test.so (build.py, -DMPC_TESTING=1): route counters behind the raw statistics and a cap on the launch grid,
// see mpc_kernel_common.h.  Must agree with the kernels' translation units.
#ifndef MPC_TESTING
#define MPC_TESTING 0
#endif
constexpr size_t kRouteWords = MPC_TESTING ? 16 : 0;

typedef unsigned long long u64;

#include "mpc_jit.h"

extern "C" {
hipError_t mpc_launch_vpc_generic(const void *, u64, const MpcVpcParams *, uint16_t *, int8_t *, u64 *, int, hipStream_t);
hipError_t mpc_launch_bdi(const void *, u64, int, uint16_t *, int8_t *, u64 *, int, hipStream_t);
hipError_t mpc_launch_fpc(const void *, u64, int, uint16_t *, int8_t *, u64 *, int, hipStream_t);
hipError_t mpc_launch_bpc(const void *, u64, int, uint16_t *, int8_t *, u64 *, int, hipStream_t);
hipError_t mpc_launch_synth(void *, u64, unsigned, int, u64, u64, const uint32_t *, hipStream_t);
hipError_t mpc_launch_read_probe(const void *, u64, uint32_t *, int, hipStream_t);
hipError_t mpc_launch_vpc_lane(const void *, u64, const MpcVpcParams *, uint16_t *, int8_t *, u64 *, int, hipStream_t);
hipError_t mpc_launch_vpc_lane_jit(hipFunction_t, hipFunction_t, const void *, u64, const MpcVpcParams *, uint16_t *, int8_t *, u64 *, int,
                                   hipStream_t);
size_t mpc_vpc_lane_smem(const MpcVpcParams *);
int mpc_vpc_lane_unrolled(const MpcVpcParams *);
size_t mpc_vpc_generic_smem(const MpcVpcParams *);
}

namespace {

thread_local std::string g_create_error;

// staging: two slots, each a pinned host buffer + device buffer + stream
constexpr size_t kStageBytes = 64ull << 20;   // per slot
constexpr size_t kMiniLines = 512;            // batches up to this many lines take the small path

struct Slot {
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  uint8_t *h_in = nullptr;       // pinned
  uint8_t *d_in = nullptr;
  uint16_t *d_sizes = nullptr, *h_sizes = nullptr;
  int8_t *d_sel = nullptr, *h_sel = nullptr;
  uint16_t *user_sizes = nullptr;  // where the pending results go
  int8_t *user_sel = nullptr;
  u64 pending_lines = 0;
  bool busy = false;
};

}  // namespace

struct mpc_handle {
  int algorithm = 0;   // 0 VPC, 1 BDI, 2 FPC, 3 BPC
  int device = 0;
  int L = 0;
  int num_cus = 256;
  mpc::VpcConfig cfg;
  mpc::VpcPlan plan;
  hipStream_t stream = nullptr;
  uint32_t *d_tab = nullptr;
  uint8_t *d_gtab = nullptr;
  u64 *d_raw = nullptr;          // device raw statistics
  u64 raw_len = 0;
  std::vector<u64> extra;        // merged-in statistics (ABI layout)
  u64 stats_len = 0;
  Slot slots[2];
  bool slots_ready = false;
  size_t stage_lines = 0;
  // small batches (the per-line CompressLine of the reference's interface above all): one pinned,
  // device-visible buffer the kernel reads the lines from and writes the results to directly --
  // no staging copies, no 64 MiB slots; a call is one launch and one stream synchronisation
  uint8_t *mini = nullptr;       // [kMiniLines * L] lines | [kMiniLines] uint16 sizes | [kMiniLines] int8 clusters
  mpcjit::Kernels jit;           // VPC: the unrolled kernels of a sequence without a built-in instantiation (mpc_jit.h)
  std::string error;
};

namespace {

int set_err(mpc_handle *h, int code, const std::string &msg)
{
  if (h) h->error = msg; else g_create_error = msg;
  return code;
}

#define HIPCHK(h, call)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return set_err((h), MPC_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));        \
  } while (0)

int pick_device(int device, int *out, int *cus)
{
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = "no HIP device available (libmpc_hip has no CPU fallback)";
    return MPC_E_NODEVICE;
  }
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  if (device >= n) {
    g_create_error = "device ordinal out of range";
    return MPC_E_INVAL;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_create_error = "hipSetDevice failed";
    return MPC_E_NODEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) *cus = prop.multiProcessorCount;
  *out = device;
  return MPC_OK;
}

int finish_create(mpc_handle *h)
{
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
    return set_err(nullptr, MPC_E_NODEVICE, "hipStreamCreate failed");
  if (hipMalloc(&h->d_raw, (h->raw_len + kRouteWords) * sizeof(u64)) != hipSuccess)
    return set_err(nullptr, MPC_E_NOMEM, "hipMalloc(stats) failed");
  if (hipMemsetAsync(h->d_raw, 0, (h->raw_len + kRouteWords) * sizeof(u64), h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess)
    return set_err(nullptr, MPC_E_NODEVICE, "hipMemset(stats) failed");
  h->extra.assign(h->stats_len, 0);
  return MPC_OK;
}

int create_vpc_from_text(const std::string &text, int device, mpc_handle **out)
{
  if (!out) return MPC_E_INVAL;
  *out = nullptr;
  mpc_handle *h = new (std::nothrow) mpc_handle();
  if (!h) return MPC_E_NOMEM;
  std::string err;
  int rc = mpc::parse_vpc_config(text, h->cfg, err);
  if (rc != 0) {
    g_create_error = err;
    delete h;
    return rc;
  }
  mpc::build_vpc_plan(h->cfg, h->plan);
  h->algorithm = 0;
  h->L = h->cfg.L;
  const int K = h->cfg.M + 1;
  h->raw_len = mpc_vpc_raw_len(K, h->cfg.hist_bins);
  h->stats_len = 3ull + 6ull * K + (u64)K * h->cfg.hist_bins;
  rc = pick_device(device, &h->device, &h->num_cus);
  if (rc == MPC_OK) {
    const size_t tb = h->plan.tab.size() * sizeof(uint32_t), gb = h->plan.gtab.size();
    if (hipMalloc(&h->d_tab, tb) != hipSuccess || hipMalloc(&h->d_gtab, gb ? gb : 16) != hipSuccess ||
        hipMemcpy(h->d_tab, h->plan.tab.data(), tb, hipMemcpyHostToDevice) != hipSuccess ||
        (gb && hipMemcpy(h->d_gtab, h->plan.gtab.data(), gb, hipMemcpyHostToDevice) != hipSuccess)) {
      g_create_error = "hipMalloc/hipMemcpy of the predictor tables failed";
      rc = MPC_E_NOMEM;
    }
  }
  if (rc == MPC_OK) {
    h->plan.params.tab = h->d_tab;
    h->plan.params.gtab = h->d_gtab;
    // a module sequence without a built-in unrolled instantiation: compile it now (mpc_jit.h); when that is not
    // possible the run-time module loop takes the configuration, and says so
    if (mpcjit::eligible(h->plan.params, h->plan.fast)) {
      std::string why;
      (void)hipSetDevice(h->device);
      if (!mpcjit::build(h->plan.params, MPC_TESTING, h->jit, why, h->plan.gtab.data()))
        std::fprintf(stderr, "libmpc_hip: module sequence [%s] runs the %s: %s\n", mpcjit::kinds_of(h->plan.params).c_str(),
                     (h->plan.params.planes_differ || h->plan.params.gather_unrolled || h->plan.params.wshift_unrolled) ? "generic kernel (some hundred times slower)" : "run-time module loop (several times slower)",
                     why.c_str());
    }
    if (h->plan.fast && (h->plan.params.planes_differ || h->plan.params.gather_unrolled || h->plan.params.wshift_unrolled) && !h->jit.mod) {
      // scan tables of different sizes, base tables that are not windowed: only a kernel compiled at creation evaluates them
      // on the fast path
      h->plan.fast = false;
      h->plan.why_generic = h->plan.params.planes_differ ? "scan tables of different sizes, and the kernel for them could not be compiled at creation"
                                                         : h->plan.params.gather_unrolled ? "BaseIndexTable is not windowed (own/previous dword), and the kernel for it could not be compiled at creation"
                                                         : "WeightTable uses more than two shift distances, and the kernel for it could not be compiled at creation";
    }
    // the statistics accumulators of a workgroup live in LDS
    const size_t smem = h->jit.mod ? mpc_vpc_lane_ring_plan(&h->plan.params, nullptr, nullptr)
                        : h->plan.fast ? mpc_vpc_lane_smem(&h->plan.params) : mpc_vpc_generic_smem(&h->plan.params);
    if (smem > 160 * 1024) {
      g_create_error = "histogram does not fit the 160 KiB LDS (too many clusters x bins)";
      rc = MPC_E_INVAL;
    }
  }
  if (rc == MPC_OK) rc = finish_create(h);
  if (rc != MPC_OK) {
    mpc_destroy(h);
    return rc;
  }
  *out = h;
  return MPC_OK;
}

// Staging copies (caller's buffer -> pinned slot) are memory-bandwidth work on the host: one
// thread moves ~12-24 GB/s, less than the PCIe link takes, so large copies are split over a
// few threads.
constexpr size_t kCopySlice = 8u << 20;
constexpr unsigned kCopyThreads = 4;

void parallel_copy(void *dst, const void *src, size_t bytes)
{
  const size_t want = (bytes + kCopySlice - 1) / kCopySlice;
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  const unsigned nt = (unsigned)std::min<size_t>(std::min<size_t>(want, kCopyThreads), hw);
  if (nt <= 1) { std::memcpy(dst, src, bytes); return; }
  const size_t per = ((bytes + nt - 1) / nt + 63) & ~(size_t)63;
  std::vector<std::thread> th;
  for (unsigned i = 0; i < nt; i++) {
    const size_t off = (size_t)i * per;
    if (off >= bytes) break;
    const size_t n = std::min(per, bytes - off);
    // (no exception may leave the library: a thread that cannot be started -- the host's thread limit -- copies here instead)
    try {
      th.emplace_back([=]() { std::memcpy((char *)dst + off, (const char *)src + off, n); });
    } catch (const std::system_error &) {
      std::memcpy((char *)dst + off, (const char *)src + off, n);
    }
  }
  for (auto &t : th) t.join();
}

// the same for page cache -> pinned slot; false on a short read / error
bool parallel_pread(int fd, void *dst, size_t bytes, u64 file_off)
{
  const size_t want = (bytes + kCopySlice - 1) / kCopySlice;
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(want, kCopyThreads), hw));
  const size_t per = ((bytes + nt - 1) / nt + 4095) & ~(size_t)4095;
  std::vector<int> ok(nt, 1);
  auto work = [&](unsigned i) {
    size_t off = (size_t)i * per;
    const size_t end = std::min(bytes, off + per);
    while (off < end) {
      const ssize_t got = pread(fd, (char *)dst + off, end - off, (off_t)(file_off + off));
      if (got <= 0) { ok[i] = 0; return; }
      off += (size_t)got;
    }
  };
  std::vector<std::thread> th;
  for (unsigned i = 1; i < nt; i++) {
    try {
      th.emplace_back(work, i);
    } catch (const std::system_error &) {
      work(i);          // (the host's thread limit: read this part here)
    }
  }
  work(0);
  for (auto &t : th) t.join();
  for (int v : ok) if (!v) return false;
  return true;
}

// Workgroups per CU of the grid-stride VPC and BDI kernels.  2-8 are resident; a grid of 32 per
// CU lets CUs that finish early pick up more work (same-box A/B against 8 per CU: VPC random
// -3.5 %, mixed -5 %, 128-byte lines -9 %, all-zero traces +3 %; BDI random -8 %, pointers -10 %).
#ifndef MPC_WG_PER_CU
#define MPC_WG_PER_CU 32
#endif
constexpr int kWgPerCu = MPC_WG_PER_CU;

int grid_for(const mpc_handle *h, u64 work_items, int block, int per_cu)
{
  u64 need = (work_items + (u64)block - 1) / (u64)block;
  u64 cap = (u64)h->num_cus * (u64)per_cu;
#if MPC_TESTING
  // test library only: MPC_TEST_GRID caps the grid so that a wave walks many groups of lines (the kernels' deferred-line
  // queues then fill and drain inside the loop even on small inputs)
  static const long test_cap = []() { const char *e = getenv("MPC_TEST_GRID"); return e ? atol(e) : 0L; }();
  if (test_cap > 0 && cap > (u64)test_cap) cap = (u64)test_cap;
#endif
  if (need < 1) need = 1;
  return (int)(need < cap ? need : cap);
}

int launch(mpc_handle *h, const void *d_lines, u64 n, uint16_t *d_sizes, int8_t *d_sel, hipStream_t s)
{
  if (n == 0) return MPC_OK;
  hipError_t e;
  if (h->algorithm == 3) {
    e = mpc_launch_bpc(d_lines, n, h->L, d_sizes, d_sel, h->d_raw, grid_for(h, n, 256, kWgPerCu), s);
  } else if (h->algorithm == 2) {
    e = mpc_launch_fpc(d_lines, n, h->L, d_sizes, d_sel, h->d_raw, grid_for(h, n, 256, kWgPerCu), s);
  } else if (h->algorithm == 1) {
    e = mpc_launch_bdi(d_lines, n, h->L, d_sizes, d_sel, h->d_raw, grid_for(h, n, 256, kWgPerCu), s);
  } else if (h->jit.mod) {
    e = mpc_launch_vpc_lane_jit(h->jit.stats, h->jit.lines, d_lines, n, &h->plan.params, d_sizes, d_sel, h->d_raw,
                                grid_for(h, n, 256, kWgPerCu), s);
  } else if (h->plan.fast) {
    e = mpc_launch_vpc_lane(d_lines, n, &h->plan.params, d_sizes, d_sel, h->d_raw, grid_for(h, n, 256, kWgPerCu), s);
  } else {
    e = mpc_launch_vpc_generic(d_lines, n, &h->plan.params, d_sizes, d_sel, h->d_raw, grid_for(h, n, 128, 8), s);
  }
  if (e != hipSuccess) return set_err(h, MPC_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return MPC_OK;
}

int ensure_slots(mpc_handle *h)
{
  if (h->slots_ready) return MPC_OK;
  h->stage_lines = kStageBytes / (size_t)h->L;
  for (int i = 0; i < 2; i++) {
    Slot &s = h->slots[i];
    HIPCHK(h, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    HIPCHK(h, hipHostMalloc((void **)&s.h_in, kStageBytes, hipHostMallocDefault));
    HIPCHK(h, hipMalloc((void **)&s.d_in, kStageBytes));
    HIPCHK(h, hipMalloc((void **)&s.d_sizes, h->stage_lines * sizeof(uint16_t)));
    HIPCHK(h, hipMalloc((void **)&s.d_sel, h->stage_lines));
    HIPCHK(h, hipHostMalloc((void **)&s.h_sizes, h->stage_lines * sizeof(uint16_t), hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void **)&s.h_sel, h->stage_lines, hipHostMallocDefault));
  }
  h->slots_ready = true;
  return MPC_OK;
}

int ensure_mini(mpc_handle *h)
{
  if (h->mini) return MPC_OK;
  const size_t bytes = kMiniLines * ((size_t)h->L + sizeof(uint16_t) + 1);
  HIPCHK(h, hipHostMalloc((void **)&h->mini, bytes, hipHostMallocDefault));
  return MPC_OK;
}

// n <= kMiniLines lines, evaluated in place from pinned host memory on the handle's own stream
int compress_small(mpc_handle *h, const uint8_t *lines, uint64_t n, uint16_t *sizes, int8_t *sel)
{
  int rc = ensure_mini(h);
  if (rc != MPC_OK) return rc;
  uint8_t *in = h->mini;
  uint16_t *out_sizes = reinterpret_cast<uint16_t *>(h->mini + kMiniLines * (size_t)h->L);
  int8_t *out_sel = reinterpret_cast<int8_t *>(out_sizes + kMiniLines);
  std::memcpy(in, lines, (size_t)(n * (uint64_t)h->L));
  rc = launch(h, in, n, sizes ? out_sizes : nullptr, sel ? out_sel : nullptr, h->stream);
  if (rc != MPC_OK) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));   // (polling hipStreamQuery instead was slower: 46 k vs 58 k lines/s)
  if (sizes) std::memcpy(sizes, out_sizes, (size_t)n * sizeof(uint16_t));
  if (sel) std::memcpy(sel, out_sel, (size_t)n);
  return MPC_OK;
}

// wait for a slot's in-flight chunk and hand its per-line results to the caller
int retire(mpc_handle *h, Slot &s)
{
  if (!s.busy) return MPC_OK;
  HIPCHK(h, hipEventSynchronize(s.done));
  if (s.user_sizes) std::memcpy(s.user_sizes, s.h_sizes, s.pending_lines * sizeof(uint16_t));
  if (s.user_sel) std::memcpy(s.user_sel, s.h_sel, s.pending_lines);
  s.busy = false;
  return MPC_OK;
}

// submit the chunk already sitting in s.h_in
int submit(mpc_handle *h, Slot &s, u64 lines, uint16_t *user_sizes, int8_t *user_sel)
{
  HIPCHK(h, hipMemcpyAsync(s.d_in, s.h_in, lines * (u64)h->L, hipMemcpyHostToDevice, s.stream));
  int rc = launch(h, s.d_in, lines, user_sizes ? s.d_sizes : nullptr, user_sel ? s.d_sel : nullptr, s.stream);
  if (rc != MPC_OK) return rc;
  if (user_sizes) HIPCHK(h, hipMemcpyAsync(s.h_sizes, s.d_sizes, lines * sizeof(uint16_t), hipMemcpyDeviceToHost, s.stream));
  if (user_sel) HIPCHK(h, hipMemcpyAsync(s.h_sel, s.d_sel, lines, hipMemcpyDeviceToHost, s.stream));
  HIPCHK(h, hipEventRecord(s.done, s.stream));
  s.user_sizes = user_sizes;
  s.user_sel = user_sel;
  s.pending_lines = lines;
  s.busy = true;
  return MPC_OK;
}

// After an error: nothing of the failed call may be delivered later.  Wait for both slots' streams
// and forget their pending results (the caller's output pointers may be gone by the next call).
void abandon_slots(mpc_handle *h)
{
  for (int i = 0; i < 2; i++) {
    Slot &s = h->slots[i];
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    s.busy = false;
    s.user_sizes = nullptr;
    s.user_sel = nullptr;
    s.pending_lines = 0;
  }
}

int sync_all(mpc_handle *h)
{
  for (int i = 0; i < 2; i++) {
    int rc = retire(h, h->slots[i]);
    if (rc != MPC_OK) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MPC_OK;
}

// raw device statistics -> ABI vector (added into `vec`)
void derive_stats(const mpc_handle *h, const std::vector<u64> &raw, std::vector<u64> &vec)
{
  if (h->algorithm == 3) {
    // lines are not recoverable from the pattern counts: the kernel counts compressed bits per line,
    // and every line contributes exactly 33 planes = TotalWords / 33
    const u64 lines = raw[7] / 33ull;
    vec[0] += lines;
    vec[1] += lines * 8ull * (u64)h->L;
    vec[2] += raw[8];
    vec[3] += raw[7];
    for (int i = 0; i < 7; i++) vec[4 + i] += raw[i];
    return;
  }
  if (h->algorithm == 2) {
    u64 words = 0;
    for (int i = 0; i < 8; i++) {
      vec[3 + i] += raw[i];
      words += raw[i];
    }
    vec[0] += words / (u64)(h->L / 4);
    vec[1] += words * 32ull;
    vec[2] += raw[8];
    return;
  }
  if (h->algorithm == 1) {
    u64 lines = 0;
    for (int i = 0; i < 9; i++) {
      vec[3 + i] += raw[i];
      lines += raw[i];
    }
    vec[0] += lines;
    vec[1] += lines * 8ull * (u64)h->L;
    vec[2] += raw[9];
    return;
  }
  const int K = h->cfg.M + 1, B = h->cfg.hist_bins;
  const u64 uncomp = 8ull * (u64)h->L;
  for (int k = 0; k < K; k++) {
    u64 count = 0, comp = 0;
    for (int s = 0; s < B; s++) {
      const u64 c = raw[2 * K + (u64)k * B + s];
      count += c;
      comp += c * (u64)s;
      vec[3 + 6 * K + (u64)k * B + s] += c;
    }
    vec[3 + 6 * k + 0] += count;
    vec[3 + 6 * k + 1] += count * uncomp;
    vec[3 + 6 * k + 2] += comp;
    // residue statistics are kept for lines that reached checkOtherPatterns (VPC.cpp:412):
    // every cluster except AllZero (0) and AllWordSame (1)
    const int cluster = k - 1;
    const bool early = (cluster == 0) || (cluster == 1 && h->cfg.has_aws);
    vec[3 + 6 * k + 3] += early ? 0 : count;
    vec[3 + 6 * k + 4] += raw[k];
    vec[3 + 6 * k + 5] += raw[K + k];
    vec[0] += count;
    vec[1] += count * uncomp;
    vec[2] += comp;
  }
}

// ---- .npy header (format spec: magic, version, header length, python dict) ----
int parse_npy_header(FILE *f, u64 *rows, u64 *cols, u64 *data_off, std::string &err)
{
  unsigned char pre[12];
  if (fread(pre, 1, 10, f) != 10 || std::memcmp(pre, "\x93NUMPY", 6) != 0) { err = "not a .npy file"; return MPC_E_PARSE; }
  size_t hlen, off;
  if (pre[6] == 1) {
    hlen = (size_t)pre[8] | ((size_t)pre[9] << 8);
    off = 10;
  } else {
    if (fread(pre + 10, 1, 2, f) != 2) { err = "truncated .npy header"; return MPC_E_PARSE; }
    hlen = (size_t)pre[8] | ((size_t)pre[9] << 8) | ((size_t)pre[10] << 16) | ((size_t)pre[11] << 24);
    off = 12;
  }
  if (hlen > (1u << 20)) { err = "unreasonable .npy header length"; return MPC_E_PARSE; }
  std::string hdr(hlen, '\0');
  if (fread(&hdr[0], 1, hlen, f) != hlen) { err = "truncated .npy header"; return MPC_E_PARSE; }
  auto find_val = [&](const char *key) -> size_t {
    size_t p = hdr.find(key);
    if (p == std::string::npos) return p;
    p = hdr.find(':', p);
    return p == std::string::npos ? p : p + 1;
  };
  size_t p = find_val("'descr'");
  if (p == std::string::npos) { err = ".npy header has no descr"; return MPC_E_PARSE; }
  size_t q1 = hdr.find('\'', p), q2 = q1 == std::string::npos ? q1 : hdr.find('\'', q1 + 1);
  if (q2 == std::string::npos) { err = ".npy descr malformed"; return MPC_E_PARSE; }
  std::string descr = hdr.substr(q1 + 1, q2 - q1 - 1);
  if (!(descr == "|u1" || descr == "<u1" || descr == "u1" || descr == "=u1")) { err = ".npy dtype is " + descr + ", expected uint8"; return MPC_E_INVAL; }
  p = find_val("'fortran_order'");
  {
    const size_t v = p == std::string::npos ? p : hdr.find_first_not_of(' ', p);
    if (v == std::string::npos) { err = ".npy header has no fortran_order value"; return MPC_E_PARSE; }
    if (hdr.compare(v, 5, "False") != 0) { err = ".npy array must be C-order"; return MPC_E_INVAL; }
  }
  p = find_val("'shape'");
  size_t a = p == std::string::npos ? p : hdr.find('(', p), b = a == std::string::npos ? a : hdr.find(')', a);
  if (b == std::string::npos) { err = ".npy shape malformed"; return MPC_E_PARSE; }
  std::vector<u64> dims;
  const char *c = hdr.c_str() + a + 1, *e = hdr.c_str() + b;
  while (c < e) {
    while (c < e && (*c < '0' || *c > '9')) c++;
    if (c >= e) break;
    u64 v = 0;
    while (c < e && *c >= '0' && *c <= '9') v = v * 10 + (u64)(*c++ - '0');
    dims.push_back(v);
  }
  if (dims.size() != 2) { err = ".npy array must be 2-D [lines, line_size]"; return MPC_E_INVAL; }
  *rows = dims[0];
  *cols = dims[1];
  *data_off = off + hlen;
  return MPC_OK;
}

uint32_t *g_sine_dev[16] = {nullptr};   // per device float32 sine period for mpc_synth_fill

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int mpc_create_vpc_from_string(const char *text, int device, mpc_handle **out)
{
  if (!text) return MPC_E_INVAL;
  return create_vpc_from_text(text, device, out);
}

int mpc_create_vpc(const char *path, int device, mpc_handle **out)
{
  if (!path || !out) return MPC_E_INVAL;
  std::string text;
  if (!mpc::read_file(path, text)) {
    g_create_error = std::string("Invalid File! \"") + path + "\" is not valid path.";
    return MPC_E_NOENT;
  }
  return create_vpc_from_text(text, device, out);
}

int mpc_create_bdi(unsigned line_size, int device, mpc_handle **out)
{
  if (!out) return MPC_E_INVAL;
  *out = nullptr;
  // BDI.cpp:8 takes any dataLine.size(); values are 8, 4 and 2 bytes wide, so a multiple of 8
  if (line_size < 8 || line_size > MPC_MAX_LINE || (line_size % 8)) {
    g_create_error = "BDI line size must be a multiple of 8 in 8.." + std::to_string(MPC_MAX_LINE) + " bytes";
    return MPC_E_INVAL;
  }
  mpc_handle *h = new (std::nothrow) mpc_handle();
  if (!h) return MPC_E_NOMEM;
  h->algorithm = 1;
  h->L = (int)line_size;
  h->raw_len = MPC_BDI_RAW_LEN;
  h->stats_len = 12;
  int rc = pick_device(device, &h->device, &h->num_cus);
  if (rc == MPC_OK) rc = finish_create(h);
  if (rc != MPC_OK) {
    mpc_destroy(h);
    return rc;
  }
  *out = h;
  return MPC_OK;
}

int mpc_create_bpc(unsigned line_size, int device, mpc_handle **out)
{
  if (!out) return MPC_E_INVAL;
  *out = nullptr;
  // 32-bit words; a plane has one bit per delta and is held in an int32_t (BPC.cpp:53-63): 2..32 words
  if (line_size < 8 || line_size > 128 || (line_size % 4)) {
    g_create_error = "BPC line size must be a multiple of 4 in 8..128 bytes";
    return MPC_E_INVAL;
  }
  mpc_handle *h = new (std::nothrow) mpc_handle();
  if (!h) return MPC_E_NOMEM;
  h->algorithm = 3;
  h->L = (int)line_size;
  h->raw_len = MPC_BPC_RAW_LEN;
  h->stats_len = 11;
  int rc = pick_device(device, &h->device, &h->num_cus);
  if (rc == MPC_OK) rc = finish_create(h);
  if (rc != MPC_OK) {
    mpc_destroy(h);
    return rc;
  }
  *out = h;
  return MPC_OK;
}

int mpc_create_fpc(unsigned line_size, int device, mpc_handle **out)
{
  if (!out) return MPC_E_INVAL;
  *out = nullptr;
  if (line_size < 4 || line_size > MPC_MAX_LINE || (line_size % 4)) {
    g_create_error = "FPC line size must be a multiple of 4 in 4.." + std::to_string(MPC_MAX_LINE) + " bytes";
    return MPC_E_INVAL;
  }
  mpc_handle *h = new (std::nothrow) mpc_handle();
  if (!h) return MPC_E_NOMEM;
  h->algorithm = 2;
  h->L = (int)line_size;
  h->raw_len = MPC_FPC_RAW_LEN;
  h->stats_len = 11;
  int rc = pick_device(device, &h->device, &h->num_cus);
  if (rc == MPC_OK) rc = finish_create(h);
  if (rc != MPC_OK) {
    mpc_destroy(h);
    return rc;
  }
  *out = h;
  return MPC_OK;
}

void mpc_destroy(mpc_handle *h)
{
  if (!h) return;
  (void)hipSetDevice(h->device);
  for (int i = 0; i < 2; i++) {
    Slot &s = h->slots[i];
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_sizes) (void)hipFree(s.d_sizes);
    if (s.d_sel) (void)hipFree(s.d_sel);
    if (s.h_sizes) (void)hipHostFree(s.h_sizes);
    if (s.h_sel) (void)hipHostFree(s.h_sel);
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.stream) (void)hipStreamDestroy(s.stream);
  }
  if (h->stream) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamDestroy(h->stream);
  }
  if (h->mini) (void)hipHostFree(h->mini);
  mpcjit::unload(h->jit);
  if (h->d_tab) (void)hipFree(h->d_tab);
  if (h->d_gtab) (void)hipFree(h->d_gtab);
  if (h->d_raw) (void)hipFree(h->d_raw);
  delete h;
}

int mpc_get_info(const mpc_handle *h, mpc_info *info)
{
  if (!h || !info) return MPC_E_INVAL;
  info->abi_version = MPC_ABI_VERSION;
  info->algorithm = h->algorithm;
  info->line_size = h->L;
  info->num_modules = h->algorithm == 0 ? h->cfg.M : 0;
  info->num_clusters = h->algorithm == 0 ? h->cfg.M + 1 : (h->algorithm == 1 ? 9 : (h->algorithm == 2 ? 8 : 7));
  info->hist_bins = h->algorithm == 0 ? h->cfg.hist_bins : 0;
  info->kernel_path = h->algorithm == 3 ? MPC_PATH_BPC : h->algorithm == 2 ? MPC_PATH_FPC
                      : h->algorithm == 1 ? MPC_PATH_BDI : (h->plan.fast ? MPC_PATH_VPC_FAST : MPC_PATH_VPC_GENERIC);
  info->device = h->device;
  info->stats_len = h->stats_len;
  return MPC_OK;
}

const char *mpc_last_error(const mpc_handle *h) { return h ? h->error.c_str() : g_create_error.c_str(); }

// Needs no device: parses the configuration and, when its module sequence would be compiled at creation, runs that
// compilation for gfx950 (nothing is loaded).  Returns the size of the code object, 0 when the sequence is built in or
// takes the run-time loop, or a negative MPC_E_* with the compiler's log in `log`.
long long mpc_jit_compile_check(const char *json_text, char *log, size_t cap)
{
  if (log && cap) log[0] = 0;
  mpc::VpcConfig cfg;
  std::string err;
  int rc = mpc::parse_vpc_config(json_text ? json_text : "", cfg, err);
  if (rc == 0) {
    mpc::VpcPlan plan;
    mpc::build_vpc_plan(cfg, plan);
    if (!mpcjit::eligible(plan.params, plan.fast)) return 0;
    unsigned ring_cfg = 0;
    const size_t smem = mpc_vpc_lane_ring_plan(&plan.params, &ring_cfg, nullptr);
    std::string code;
    if (mpcjit::compile(mpcjit::source_of(plan.params, smem, MPC_TESTING, plan.gtab.data()), "gfx950", mpcjit::source_dir(), code, err)) return (long long)code.size();
    rc = MPC_E_HIP;
  }
  if (log && cap) {
    std::strncpy(log, err.c_str(), cap - 1);
    log[cap - 1] = 0;
  }
  return rc;
}

const char *mpc_kernel_form(const mpc_handle *h)
{
  if (!h) return "";
  if (h->algorithm != 0) return "unrolled";
  if (!h->plan.fast) return "generic";
  if (h->jit.mod) return h->jit.from_cache ? "unrolled, compiled at creation (from the cache)" : "unrolled, compiled at creation";
  if (!mpc_vpc_lane_unrolled(&h->plan.params)) return "run-time loop";
  return h->plan.params.gen_layout ? "unrolled, general layout" : "unrolled";
}

const char *mpc_path_reason(const mpc_handle *h)
{
  return (h && h->algorithm == 0 && !h->plan.fast) ? h->plan.why_generic.c_str() : "";
}

int mpc_compress_batch_device(mpc_handle *h, const void *d_lines, uint64_t n, uint16_t *d_sizes, int8_t *d_sel,
                              void *hip_stream)
{
  if (!h || (!d_lines && n)) return MPC_E_INVAL;
  if (((uintptr_t)d_lines) & 15u) return set_err(h, MPC_E_INVAL, "device line buffer must be 16-byte aligned");
  HIPCHK(h, hipSetDevice(h->device));
  return launch(h, d_lines, n, d_sizes, d_sel, (hipStream_t)hip_stream);
}

int mpc_sync(mpc_handle *h)
{
  if (!h) return MPC_E_INVAL;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = sync_all(h);
  if (rc != MPC_OK) return rc;
  HIPCHK(h, hipDeviceSynchronize());   // batches may have been queued on caller streams
  return MPC_OK;
}

int mpc_compress_batch(mpc_handle *h, const uint8_t *lines, uint64_t n, uint16_t *sizes, int8_t *sel)
{
  if (!h || (!lines && n)) return MPC_E_INVAL;
  if (n == 0) return MPC_OK;
  if (n <= kMiniLines) {
    HIPCHK(h, hipSetDevice(h->device));
    return compress_small(h, lines, n, sizes, sel);
  }
  HIPCHK(h, hipSetDevice(h->device));
  int rc = ensure_slots(h);
  if (rc != MPC_OK) return rc;
  u64 done = 0;
  int which = 0;
  while (done < n) {
    Slot &s = h->slots[which];
    rc = retire(h, s);   // the slot's previous chunk (overlapped with the other slot's work)
    if (rc != MPC_OK) break;
    const u64 take = (n - done) < (u64)h->stage_lines ? (n - done) : (u64)h->stage_lines;
    parallel_copy(s.h_in, lines + done * (u64)h->L, (size_t)(take * (u64)h->L));
    rc = submit(h, s, take, sizes ? sizes + done : nullptr, sel ? sel + done : nullptr);
    if (rc != MPC_OK) break;
    done += take;
    which ^= 1;
  }
  if (rc == MPC_OK) rc = sync_all(h);
  if (rc != MPC_OK) abandon_slots(h);
  return rc;
}

int mpc_npy_shape(const char *path, uint64_t *rows, uint64_t *cols)
{
  if (!path || !rows || !cols) return MPC_E_INVAL;
  FILE *f = fopen(path, "rb");
  if (!f) { g_create_error = std::string("cannot open ") + path; return MPC_E_NOENT; }
  u64 r, c, off;
  std::string err;
  int rc = parse_npy_header(f, &r, &c, &off, err);
  fclose(f);
  if (rc != MPC_OK) { g_create_error = err; return rc; }
  *rows = r;
  *cols = c;
  return MPC_OK;
}

int mpc_compress_npy(mpc_handle *h, const char *path, uint64_t first_row, uint64_t n_rows, int skip_last_row,
                     uint64_t *rows_done)
{
  if (!h || !path) return MPC_E_INVAL;
  if (rows_done) *rows_done = 0;
  FILE *f = fopen(path, "rb");
  if (!f) return set_err(h, MPC_E_NOENT, std::string("cannot open ") + path);
  u64 rows, cols, off;
  std::string err;
  int rc = parse_npy_header(f, &rows, &cols, &off, err);
  if (rc != MPC_OK) { fclose(f); return set_err(h, rc, err); }
  if (cols != (u64)h->L) {
    fclose(f);
    return set_err(h, MPC_E_INVAL, "trace line size " + std::to_string(cols) + " differs from the evaluator's " + std::to_string(h->L));
  }
  // the reference driver drops the final row (LoaderNPY.cpp:28-32 + main.cpp:240)
  u64 usable = (skip_last_row && rows > 0) ? rows - 1 : rows;
  u64 begin = first_row < usable ? first_row : usable;
  u64 end = (n_rows > usable - begin) ? usable : begin + n_rows;
  if (hipSetDevice(h->device) != hipSuccess) { fclose(f); return set_err(h, MPC_E_HIP, "hipSetDevice failed"); }
  rc = ensure_slots(h);
  if (rc != MPC_OK) { fclose(f); return rc; }
  const int fd = fileno(f);
  u64 done = begin;
  int which = 0;
  while (done < end) {
    Slot &s = h->slots[which];
    rc = retire(h, s);
    if (rc != MPC_OK) break;
    const u64 take = (end - done) < (u64)h->stage_lines ? (end - done) : (u64)h->stage_lines;
    if (!parallel_pread(fd, s.h_in, (size_t)(take * cols), off + done * cols)) { rc = set_err(h, MPC_E_PARSE, "short read: .npy file is truncated"); break; }
    rc = submit(h, s, take, nullptr, nullptr);
    if (rc != MPC_OK) break;
    done += take;
    which ^= 1;
  }
  fclose(f);
  if (rc == MPC_OK) rc = sync_all(h);
  if (rc != MPC_OK) abandon_slots(h);
  if (rc == MPC_OK && rows_done) *rows_done = end - begin;
  return rc;
}

namespace {

constexpr int kLogKeys = 17, kLogRecordHeader = 62;

// validates the file header of a GPGPU-Sim trace (LoaderGPGPU.cpp:93-119)
int log_open(const char *path, FILE **out, std::string &err)
{
  FILE *f = fopen(path, "rb");
  if (!f) { err = std::string("Failed to open a file. Check the path of the file: ") + path; return MPC_E_NOENT; }
  unsigned char hdr[1 + 7 * kLogKeys];
  if (fread(hdr, 1, sizeof(hdr), f) != sizeof(hdr) || hdr[0] != kLogKeys) {
    fclose(f);
    err = "The header of the GPGPU-sim trace file is not valid.";
    return MPC_E_PARSE;
  }
  *out = f;
  return MPC_OK;
}

// one request header; false at the end of the file (or inside an incomplete header)
bool log_next(FILE *f, uint32_t *req_type, uint32_t *req_size)
{
  unsigned char h[kLogRecordHeader];
  if (fread(h, 1, sizeof(h), f) != sizeof(h)) return false;
  std::memcpy(req_type, h + 38, 4);
  std::memcpy(req_size, h + 58, 4);
  return true;
}

}  // namespace

int mpc_gpgpusim_log_line_size(const char *log_path, uint32_t *line_size)
{
  if (!log_path || !line_size) return MPC_E_INVAL;
  *line_size = 0;
  FILE *f = nullptr;
  std::string err;
  int rc = log_open(log_path, &f, err);
  if (rc != MPC_OK) return set_err(nullptr, rc, err);
  uint32_t t = 0, sz = 0;
  if (log_next(f, &t, &sz)) *line_size = sz;
  fclose(f);
  return MPC_OK;
}

int mpc_compress_gpgpusim_log(mpc_handle *h, const char *log_path, uint64_t *requests_read, uint64_t *lines_done)
{
  if (!h || !log_path) return MPC_E_INVAL;
  if (requests_read) *requests_read = 0;
  if (lines_done) *lines_done = 0;
  // the file is mapped and walked in memory (per-request stdio calls cap the rate at ~35 M requests/s)
  const int fd = open(log_path, O_RDONLY);
  if (fd < 0) return set_err(h, MPC_E_NOENT, std::string("Failed to open a file. Check the path of the file: ") + log_path);
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return set_err(h, MPC_E_NOENT, std::string("cannot stat ") + log_path); }
  const u64 size = (u64)st.st_size;
  constexpr u64 kFileHeader = 1 + 7 * kLogKeys;
  const unsigned char *base = nullptr;
  if (size > 0) {
    void *m = mmap(nullptr, (size_t)size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); return set_err(h, MPC_E_NOMEM, std::string("cannot map ") + log_path); }
    base = static_cast<const unsigned char *>(m);
    (void)madvise(m, (size_t)size, MADV_SEQUENTIAL);
  }
  close(fd);
  auto unmap = [&]() { if (base) munmap(const_cast<unsigned char *>(base), (size_t)size); };
  if (size < kFileHeader || base[0] != kLogKeys) {
    unmap();
    return set_err(h, MPC_E_PARSE, "The header of the GPGPU-sim trace file is not valid.");
  }
  if (hipSetDevice(h->device) != hipSuccess) { unmap(); return set_err(h, MPC_E_HIP, "hipSetDevice failed"); }
  int rc = ensure_slots(h);
  if (rc != MPC_OK) { unmap(); return rc; }
  const u64 L = (u64)h->L;
  u64 requests = 0, lines = 0, fill = 0, pos = kFileHeader;
  bool first = true;
  int which = 0;
  rc = retire(h, h->slots[which]);
  while (rc == MPC_OK && pos + kLogRecordHeader <= size) {
    uint32_t req_type, req_size;
    std::memcpy(&req_type, base + pos + 38, 4);
    std::memcpy(&req_size, base + pos + 58, 4);
    if (first && req_size != L) {
      rc = set_err(h, MPC_E_INVAL, "trace line size " + std::to_string(req_size) + " differs from the evaluator's " + std::to_string(L));
      break;
    }
    first = false;
    const u64 next = pos + kLogRecordHeader + (u64)req_size;
    if (next > size) break;                                      // incomplete trailing request
    if (req_type == 0u || req_type == 4u) {                      // GLOBAL_ACC_R, GLOBAL_ACC_W
      if (req_size != L) {
        rc = set_err(h, MPC_E_INVAL, "the GPGPU-sim trace mixes request sizes (" + std::to_string(req_size) + " after " + std::to_string(L) + " bytes)");
        break;
      }
      std::memcpy(h->slots[which].h_in + fill * L, base + pos + kLogRecordHeader, (size_t)L);
      fill++;
    }
    pos = next;
    requests++;
    if (fill == (u64)h->stage_lines) {
      rc = submit(h, h->slots[which], fill, nullptr, nullptr);
      if (rc != MPC_OK) break;
      lines += fill;
      fill = 0;
      which ^= 1;
      rc = retire(h, h->slots[which]);
    }
  }
  if (rc == MPC_OK && fill) {
    rc = submit(h, h->slots[which], fill, nullptr, nullptr);
    if (rc == MPC_OK) lines += fill;
  }
  if (rc == MPC_OK) rc = sync_all(h);
  if (rc != MPC_OK) abandon_slots(h);
  unmap();
  if (rc == MPC_OK) {
    if (requests_read) *requests_read = requests;
    if (lines_done) *lines_done = lines;
  }
  return rc;
}

int mpc_stats_len(const mpc_handle *h, uint64_t *len)
{
  if (!h || !len) return MPC_E_INVAL;
  *len = h->stats_len;
  return MPC_OK;
}

int mpc_stats_get(mpc_handle *h, uint64_t *vec, size_t n)
{
  if (!h || !vec || n < h->stats_len) return MPC_E_INVAL;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = sync_all(h);
  if (rc != MPC_OK) return rc;
  HIPCHK(h, hipDeviceSynchronize());   // callers may have used their own streams
  std::vector<u64> raw(h->raw_len);
  HIPCHK(h, hipMemcpy(raw.data(), h->d_raw, h->raw_len * sizeof(u64), hipMemcpyDeviceToHost));
  std::vector<u64> out(h->extra);
  derive_stats(h, raw, out);
  std::memcpy(vec, out.data(), h->stats_len * sizeof(u64));
  return MPC_OK;
}

int mpc_stats_raw_len(const mpc_handle *h, uint64_t *raw_len)
{
  if (!h || !raw_len) return MPC_E_INVAL;
  *raw_len = h->raw_len;
  return MPC_OK;
}

int mpc_stats_copy_raw_device(mpc_handle *h, void *d_dst, void *hip_stream)
{
  if (!h || !d_dst) return MPC_E_INVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(d_dst, h->d_raw, h->raw_len * sizeof(u64), hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(hip_stream)));
  return MPC_OK;
}

int mpc_stats_from_raw(const mpc_handle *h, const uint64_t *raw, size_t raw_len, uint64_t *vec, size_t n)
{
  if (!h || !raw || !vec || raw_len != h->raw_len || n < h->stats_len) return MPC_E_INVAL;
  std::vector<u64> r(raw, raw + raw_len), out(h->stats_len, 0);
  derive_stats(h, r, out);
  std::memcpy(vec, out.data(), h->stats_len * sizeof(u64));
  return MPC_OK;
}

int mpc_stats_merge(mpc_handle *h, const uint64_t *vec, size_t n)
{
  if (!h || !vec || n != h->stats_len) return MPC_E_INVAL;
  for (size_t i = 0; i < n; i++) h->extra[i] += vec[i];
  return MPC_OK;
}

int mpc_stats_reset(mpc_handle *h)
{
  if (!h) return MPC_E_INVAL;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = sync_all(h);
  if (rc != MPC_OK) return rc;
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemset(h->d_raw, 0, (h->raw_len + kRouteWords) * sizeof(u64)));
  h->extra.assign(h->stats_len, 0);
  return MPC_OK;
}

int mpc_stats_set(mpc_handle *h, const uint64_t *vec, size_t n)
{
  if (!h || !vec || n != h->stats_len) return MPC_E_INVAL;
  int rc = mpc_stats_reset(h);
  if (rc != MPC_OK) return rc;
  for (size_t i = 0; i < n; i++) h->extra[i] = vec[i];
  return MPC_OK;
}

int mpc_config_describe(const char *json_text, char *out, size_t cap)
{
  if (!json_text || !out || cap < 2) return MPC_E_INVAL;
  mpc::VpcConfig cfg;
  std::string err;
  int rc = mpc::parse_vpc_config(json_text, cfg, err);
  std::string s;
  if (rc != 0) {
    s = "{\"error\": \"";
    for (char c : err) s += (c == '"' || c == '\\') ? '\'' : c;
    s += "\"}";
  } else {
    mpc::VpcPlan plan;
    mpc::build_vpc_plan(cfg, plan);
    if (plan.fast && (plan.params.planes_differ || plan.params.gather_unrolled || plan.params.wshift_unrolled) && !mpcjit::eligible(plan.params, true)) {      // (as mpc_create_vpc* decides)
      plan.fast = false;
      plan.why_generic = plan.params.planes_differ ? "scan tables of different sizes, and run-time compilation is not available for them"
                                                   : plan.params.gather_unrolled ? "BaseIndexTable is not windowed (own/previous dword), and run-time compilation is not available for it"
                                                   : "WeightTable uses more than two shift distances, and run-time compilation is not available for it";
    }
    s = "{\"L\": " + std::to_string(cfg.L) + ", \"M\": " + std::to_string(cfg.M) + ", \"n_pred\": " + std::to_string(cfg.n_pred) +
        ", \"has_aws\": " + (cfg.has_aws ? "true" : "false") + ", \"hist_bins\": " + std::to_string(cfg.hist_bins) + ", \"enc_bits\": [";
    for (size_t i = 0; i < cfg.enc_bits.size(); i++) s += (i ? ", " : "") + std::to_string(cfg.enc_bits[i]);
    s += "], \"path\": \"" + std::string(plan.fast ? "fast" : "generic") + "\", \"why_generic\": \"" + plan.why_generic +
         "\", \"sequence\": \"" + std::string(!plan.fast ? "" : (mpc_vpc_lane_unrolled(&plan.params) || mpcjit::eligible(plan.params, plan.fast) ? "unrolled" : "run-time loop")) +
         "\", \"compiled\": \"" + std::string(!plan.fast ? "" : mpc_vpc_lane_unrolled(&plan.params) ? "built in" :
                                            mpcjit::eligible(plan.params, plan.fast) ? "at creation" : "") +
         "\", \"general_layout\": \"" + std::string(plan.fast && plan.params.gen_layout && (mpc_vpc_lane_unrolled(&plan.params) || mpcjit::eligible(plan.params, plan.fast)) ? "yes" : "no") +
         "\", \"scan_order\": \"" + std::string(!plan.fast ? "" : (plan.params.byte_major ? "byte-major" : "plane-major")) +
         "\", \"modules\": [";
    for (int i = 0; i < cfg.M; i++) {
      const mpc::Module &m = cfg.modules[(size_t)i];
      s += (i ? ", " : "");
      s += "{\"kind\": " + std::to_string(m.kind) + ", \"pred_kind\": " + std::to_string(m.pred_kind) + ", \"root\": " + std::to_string(m.root) +
           ", \"cx\": " + (m.consecutive_xor ? "1" : "0") + ", \"table_size\": " + std::to_string(m.table_size) + ", \"shifts\": [";
      for (size_t j = 0; j < m.weight.size(); j++)
        s += (j ? "," : "") + std::to_string(m.pred_kind == mpc::PRED_WEIGHT && (int)j != m.root ? mpc::weight_shift(m.weight[j]) : 0);
      s += "]}";
    }
    s += "]}";
  }
  if (s.size() + 1 > cap) return MPC_E_NOMEM;
  std::memcpy(out, s.c_str(), s.size() + 1);
  return rc;
}

int mpc_synth_fill(void *d_lines, uint64_t n_lines, unsigned L, int kind, uint64_t first_line, uint64_t seed, void *stream)
{
  if (!d_lines || L % 8 || kind < 0 || kind > 4) return MPC_E_INVAL;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return MPC_E_NODEVICE;
  if (!g_sine_dev[dev]) {
    // float32(sin(2*pi*t/1024)) computed in double on the host: the same
    // values as cal_22-mpc_amd/traces.py:sine_table()
    float tabf[1024];
    for (int t = 0; t < 1024; t++) tabf[t] = (float)sin(2.0 * 3.14159265358979323846 * (double)t / 1024.0);
    if (hipMalloc((void **)&g_sine_dev[dev], sizeof(tabf)) != hipSuccess) return MPC_E_NOMEM;
    if (hipMemcpy(g_sine_dev[dev], tabf, sizeof(tabf), hipMemcpyHostToDevice) != hipSuccess) return MPC_E_HIP;
  }
  hipError_t e = mpc_launch_synth(d_lines, n_lines, L, kind, first_line, seed, g_sine_dev[dev], (hipStream_t)stream);
  return e == hipSuccess ? MPC_OK : MPC_E_HIP;
}

int mpc_read_bandwidth_probe(const void *d_buf, uint64_t bytes, void *stream)
{
  static uint32_t *sink[16] = {nullptr};
  int dev = 0;
  if (!d_buf || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return MPC_E_INVAL;
  if (!sink[dev] && hipMalloc((void **)&sink[dev], 64) != hipSuccess) return MPC_E_NOMEM;
  hipError_t e = mpc_launch_read_probe(d_buf, bytes, sink[dev], 256 * 8, (hipStream_t)stream);
  return e == hipSuccess ? MPC_OK : MPC_E_HIP;
}

#if MPC_TESTING
// test library only (not part of include/mpc_hip.h): the kernels' route counters since the handle was created or
// its statistics were last reset; out[i] for i < n, order of the MPC_RT_* enumeration (mpc_kernel_common.h)
int mpc_test_routes(mpc_handle *h, uint64_t *out, size_t n)
{
  if (!h || !out) return MPC_E_INVAL;
  if (hipSetDevice(h->device) != hipSuccess) return MPC_E_HIP;
  HIPCHK(h, hipDeviceSynchronize());
  uint64_t tmp[kRouteWords];
  HIPCHK(h, hipMemcpy(tmp, h->d_raw + h->raw_len, sizeof(tmp), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; i++) out[i] = i < kRouteWords ? tmp[i] : 0;
  return MPC_OK;
}
#endif

}  // extern "C"
