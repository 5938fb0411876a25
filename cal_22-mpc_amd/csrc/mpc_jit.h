// mpc_jit.h -- run-time compilation of the unrolled lane kernel for ONE more module sequence (host side, used by
// mpc_capi.hip when a handle is created).
//
// The library ships the unrolled kernels for 13 module sequences (mpc_vpc_lane.hip, MPC_LANE_SEQUENCES); any other
// sequence of OneBase / ConsecutiveBase / DiffBase / WeightBase modules used to fall back to the run-time module loop
// at about 0.3 of the HBM peak -- as did every configuration with the byte-major scan order, for which
// the library has no unrolled kernel at all (compiled here with -DMPC_JIT_BM=1).  The group code is a template over the sequence, so the missing
// instantiation is compiled when the configuration is loaded: hiprtc (the ROCm run-time compiler; in a helper process,
// mpc_jitc, so that the library neither links it nor shares a process with it) compiles mpc_vpc_lane.hip -- the very source the built-in kernels come from, read
// from the csrc/ directory next to the library -- with the sequence as template arguments, the code object is loaded
// with hipModuleLoadData and kept in the handle; a copy goes to a cache directory keyed by a hash of the sources, the
// options and the compiler version, so that the next process skips the compilation (about 3 s).  Nothing here is a
// fallback path of the product: when hiprtc, the sources or the compilation are not available, the configuration runs
// the run-time module loop as before and a line on stderr says why.
//
//   MPC_JIT=0            never compile at run time
//   MPC_JITC=PATH        the helper program (default: mpc_jitc next to the library; without one hiprtc is dlopen'ed here)
//   MPC_JIT_SRC=DIR      the kernel sources (default: <directory of libmpc_hip*.so>/csrc)
//   MPC_JIT_MAX_MODULES=N  longest module sequence that is compiled (default 12, at most 16)
//   MPC_JIT_DEBUG=1      say on stderr where the compilation ran
//   MPC_JIT_CACHE=DIR    code object cache (default: $XDG_CACHE_HOME/mpc_hip or ~/.cache/mpc_hip); "" = no cache
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "mpc_device.h"

extern "C" size_t mpc_vpc_lane_ring_plan(const MpcVpcParams *P, unsigned *ring_cfg, int *wpb);
extern "C" int mpc_vpc_lane_unrolled(const MpcVpcParams *P);

namespace mpcjit {

// Sequences of more than 12 prediction modules stay on the run-time loop (MPC_JIT_MAX_MODULES overrides, up to 16): the
// unrolled code still wins there (tools/dev/seq_len.py: 12 modules 0.42 against 0.22 of the peak on random data, 14 modules
// 0.34 against 0.19) but spills more and more registers (84 VGPRs at 12 modules) and takes longer to compile.
inline int max_modules()
{
  const char *e = std::getenv("MPC_JIT_MAX_MODULES");
  const int v = e ? std::atoi(e) : 12;
  return v < 1 ? 1 : (v > 16 ? 16 : v);
}

struct Kernels {
  hipModule_t mod = nullptr;
  hipFunction_t stats = nullptr;    // statistics only
  hipFunction_t lines = nullptr;    // + per-line outputs
  bool from_cache = false;
};

// Loaded modules belong to the process, not to a handle: they are kept (with the code object image they were loaded from)
// until the process ends and shared by every handle of the same configuration shape on the same device.  A handle's
// Kernels are borrowed references; nothing is ever unloaded (a module is ~100 KB of device memory).
struct Loaded {
  hipModule_t mod = nullptr;
  hipFunction_t stats = nullptr, lines = nullptr;
  std::string image;      // the code object, alive as long as the module
};
inline std::mutex &loaded_mutex() { static std::mutex m; return m; }
inline std::map<std::string, Loaded *> &loaded_modules() { static std::map<std::string, Loaded *> m; return m; }

inline void unload(Kernels &k) { k = Kernels(); }

// the sequence as the kernels' template arguments (lane-kernel kinds: MPC_FK_* | 8 / 16 for periodic tables)
inline std::string kinds_of(const MpcVpcParams &P)
{
  std::string s;
  for (int q = 0; q < P.n_pred; q++) {
    const MpcFastModule &f = P.fm[q];
    const int kind = f.kind | (f.prev_word == 1 ? 8 : (f.prev_word == 2 ? 16 : 0));
    s += (q ? ", " : "") + std::to_string(kind);
  }
  return s;
}

inline std::string wgens_of(const MpcVpcParams &P)
{
  std::string s;
  for (int q = 0; q < P.n_pred; q++) s += (q ? ", " : "") + std::to_string(P.fm[q].wgen ? 1 : 0);
  return s;
}

inline std::string gathers_of(const MpcVpcParams &P)
{
  std::string s;
  for (int q = 0; q < P.n_pred; q++) s += (q ? ", " : "") + std::to_string(P.fm[q].gather ? 1 : 0);
  return s;
}

inline std::string roots_of(const MpcVpcParams &P)
{
  std::string s;
  for (int q = 0; q < P.n_pred; q++) s += (q ? ", " : "") + std::to_string(P.anyroot_unrolled ? P.fm[q].root : 0);
  return s;
}

// a fast-path configuration whose sequence has no built-in instantiation but could have one
inline bool eligible(const MpcVpcParams &P, bool fast)
{
  const char *env = std::getenv("MPC_JIT");
  if (env && std::strcmp(env, "0") == 0) return false;
  if (!fast || (P.runtime_only && !P.bm_unrolled && !P.anyroot_unrolled && !P.planes_differ && !P.gather_unrolled && !P.wshift_unrolled) || P.n_pred < 1 || P.n_pred > max_modules()) return false;
  if (!(P.L == 32 || P.L == 64 || P.L == 128)) return false;
  if (mpc_vpc_lane_unrolled(&P)) return false;            // built in (never the byte-major order)
  unsigned ring_cfg = 0;
  return mpc_vpc_lane_ring_plan(&P, &ring_cfg, nullptr) != 0;       // the rings fit beside the histogram (with some workgroup size)
}

namespace detail {

inline std::string dir_of_this_library()
{
  Dl_info info;
  if (!dladdr(reinterpret_cast<const void *>(&mpc_vpc_lane_ring_plan), &info) || !info.dli_fname) return "";
  std::string p = info.dli_fname;
  const size_t slash = p.rfind('/');
  return slash == std::string::npos ? "." : p.substr(0, slash);
}

inline bool read_file(const std::string &path, std::string &out)
{
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}

inline unsigned long long fnv1a(const std::string &s, unsigned long long h = 1469598103934665603ull)
{
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

inline std::string cache_dir()
{
  const char *e = std::getenv("MPC_JIT_CACHE");
  if (e) return e;                                         // ("" switches the cache off)
  const char *x = std::getenv("XDG_CACHE_HOME");
  if (x && *x) return std::string(x) + "/mpc_hip";
  const char *h = std::getenv("HOME");
  if (h && *h) return std::string(h) + "/.cache/mpc_hip";
  return "";
}

inline void mkdirs(const std::string &dir)
{
  for (size_t i = 1; i <= dir.size(); i++)
    if (i == dir.size() || dir[i] == '/') (void)::mkdir(dir.substr(0, i).c_str(), 0755);
}

struct Rtc {
  void *lib = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
  decltype(&hiprtcVersion) version = nullptr;
  bool ok() const { return create && compile && log_size && log && code_size && code && destroy; }
};

inline Rtc &rtc()
{
  static Rtc r = []() {
    Rtc t;
    for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      t.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (t.lib) break;
    }
    if (!t.lib) return t;
#define MPC_RTC_SYM(field, sym) t.field = reinterpret_cast<decltype(t.field)>(dlsym(t.lib, #sym))
    MPC_RTC_SYM(create, hiprtcCreateProgram);
    MPC_RTC_SYM(compile, hiprtcCompileProgram);
    MPC_RTC_SYM(log_size, hiprtcGetProgramLogSize);
    MPC_RTC_SYM(log, hiprtcGetProgramLog);
    MPC_RTC_SYM(code_size, hiprtcGetCodeSize);
    MPC_RTC_SYM(code, hiprtcGetCode);
    MPC_RTC_SYM(destroy, hiprtcDestroyProgram);
    MPC_RTC_SYM(version, hiprtcVersion);
#undef MPC_RTC_SYM
    return t;
  }();
  return r;
}

}  // namespace detail

// the translation unit handed to hiprtc: the lane kernel's device code and two kernels of the sequence
// gtab: the HOST copy of the generic-path tables (VpcPlan::gtab: every module's BaseIndexTable at gm[q].off_base)
inline std::string bases_of(const MpcVpcParams &P, const unsigned char *gtab)
{
  std::string s;
  for (int q = 0; q < P.n_pred; q++) {
    s += q ? ", {" : "{";
    for (int i = 0; i < P.L; i++) {
      const bool has = P.fm[q].gather && gtab && i != P.fm[q].root;
      s += (i ? "," : "") + std::to_string(has ? (int)gtab[P.gm[q].off_base + i] : -1);
    }
    s += "}";
  }
  return s;
}

// every module's shift distance per byte (VpcPlan::gtab at gm[q].off_shift, clamped to +-8); 99: the predicted byte is 0
inline std::string shifts_of(const MpcVpcParams &P, const unsigned char *gtab)
{
  std::string s;
  for (int q = 0; q < P.n_pred; q++) {
    s += q ? ", {" : "{";
    for (int i = 0; i < P.L; i++) {
      int v = 99;
      if (P.fm[q].wgen && gtab && i != P.fm[q].root) {
        v = (int)(signed char)gtab[P.gm[q].off_shift + i];
        if (v >= 8 || v <= -8) v = 99;
      }
      s += (i ? "," : "") + std::to_string(v);
    }
    s += "}";
  }
  return s;
}

inline std::string source_of(const MpcVpcParams &P, size_t smem_bytes, int testing, const unsigned char *gtab = nullptr)
{
  const int W = P.L / 4;
  int wpb = 0;
  (void)mpc_vpc_lane_ring_plan(&P, nullptr, &wpb);
  const std::string targs = std::to_string(W) + ", OUT_, " + (P.gen_layout ? "true" : "false") + ", " + kinds_of(P);
  std::ostringstream s;
  s << "#define MPC_LANE_JIT 1\n"
    << "#define MPC_JIT_BM " << (P.bm_unrolled ? 1 : 0) << "\n"
    << "#define MPC_JIT_ANYROOT " << (P.anyroot_unrolled ? 1 : 0) << "\n"
    << "#define MPC_JIT_PLANES " << (P.planes_differ ? 1 : 0) << "\n"
    << "#define MPC_JIT_WSHIFT " << (P.wshift_unrolled ? 1 : 0) << "\n"
    << "#define MPC_JIT_WGENS " << wgens_of(P) << "\n"
    << "#define MPC_JIT_SHIFTS " << (P.wshift_unrolled ? shifts_of(P, gtab) : std::string("{0}")) << "\n"
    << "#define MPC_JIT_GATHER " << (P.gather_unrolled ? 1 : 0) << "\n"
    << "#define MPC_JIT_GATHERS " << gathers_of(P) << "\n"
    << "#define MPC_JIT_BASES " << (P.gather_unrolled ? bases_of(P, gtab) : std::string("{0}")) << "\n"
    << "#define MPC_JIT_ROOTS " << roots_of(P) << "\n"
    << "#define MPC_TESTING " << testing << "\n"
    << "#define " << (P.L <= 32 ? "MPC_LANE_WAVES_32 " : P.L <= 64 ? "MPC_LANE_WAVES " : "MPC_LANE_WAVES_128 ") << wpb << "      /* waves per workgroup of the LDS plan */\n"
    << "#include \"mpc_vpc_lane.hip\"\n"
    << "#define MPC_JIT_KERNEL(NAME, OUT_) \\\n"
    << "extern \"C\" __global__ void MPC_LANE_BOUNDS(" << W << ", " << P.n_pred << ") \\\n"
    << "NAME(const uint4 *__restrict__ lines, u32 n_lines, u64 first_line, MpcVpcParams P, uint16_t *__restrict__ sizes_out, \\\n"
    << "     int8_t *__restrict__ sel_out, u64 *gstats, u32 ring_cfg) \\\n"
    << "{ \\\n"
    << "  /* the launch's LDS as ONE static array (it sits at LDS address 0, where the ring code expects it) */ \\\n"
    << "  __shared__ __attribute__((aligned(1024))) unsigned char smem[" << smem_bytes << "]; \\\n"
    << "  if ((u32)(uintptr_t)smem != 0u) __builtin_trap(); \\\n"
    << "  vpc_lane_body<" << targs << ">(smem, lines, n_lines, first_line, P, sizes_out, sel_out, gstats, ring_cfg); \\\n"
    << "}\n"
    << "MPC_JIT_KERNEL(mpc_jit_stats, false)\n"
    << "MPC_JIT_KERNEL(mpc_jit_lines, true)\n";
  return s.str();
}

inline std::string source_dir()
{
  const char *src_env = std::getenv("MPC_JIT_SRC");
  return src_env ? std::string(src_env) : detail::dir_of_this_library() + "/csrc";
}

// The compiler runs in a process of its own (mpc_jitc, built next to the library; MPC_JITC names another one): a host
// application may carry its own copy of the ROCm compiler libraries (a PyTorch wheel bundles libhiprtc / libamd_comgr),
// two LLVMs in one process are asking for trouble, and a fatal compiler error must not take the application down.
// Returns 1 = code object read back, 0 = the helper ran and failed (why), -1 = no helper to start.
inline int compile_in_helper(const std::string &source, const std::string &arch, const std::string &src_dir, std::string &code,
                             std::string &why)
{
  const char *env = std::getenv("MPC_JITC");
  const std::string helper = env ? std::string(env) : detail::dir_of_this_library() + "/mpc_jitc";
  if (::access(helper.c_str(), X_OK) != 0) return -1;
  const char *tmp_env = std::getenv("TMPDIR");
  static std::atomic<unsigned> serial{0};                 // (handles may be created from several threads)
  char stem[512];
  std::snprintf(stem, sizeof stem, "%s/mpc_jit_%ld_%u_%llx", (tmp_env && *tmp_env) ? tmp_env : "/tmp", (long)getpid(),
                serial.fetch_add(1u), detail::fnv1a(source) ^ (unsigned long long)::time(nullptr));
  const std::string in = std::string(stem) + ".hip", outp = std::string(stem) + ".hsaco", logp = std::string(stem) + ".log";
  {
    std::ofstream f(in, std::ios::binary);
    f << source;
    if (!f) { why = "cannot write " + in; return 0; }
  }
  posix_spawn_file_actions_t fa;
  posix_spawn_file_actions_init(&fa);
  posix_spawn_file_actions_addopen(&fa, 2, logp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
  posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);
  char *argv[] = {const_cast<char *>(helper.c_str()), const_cast<char *>(in.c_str()), const_cast<char *>(arch.c_str()),
                  const_cast<char *>(src_dir.c_str()), const_cast<char *>(outp.c_str()), nullptr};
  // the child gets this process's environment minus preloaded libraries (a profiler's, a sanitizer's): it only compiles
  std::vector<std::string> keep;
  for (char **e = environ; e && *e; e++)
    if (std::strncmp(*e, "LD_PRELOAD=", 11) != 0) keep.push_back(*e);
  std::vector<char *> envp;
  for (std::string &e : keep) envp.push_back(&e[0]);
  envp.push_back(nullptr);
  pid_t pid = 0;
  const int rc = posix_spawn(&pid, helper.c_str(), &fa, nullptr, argv, envp.data());
  posix_spawn_file_actions_destroy(&fa);
  int result = 0;
  if (rc != 0) {
    why = "cannot start " + helper;
    result = -1;
  } else {
    int status = 0;
    bool done = false;
    for (int waited = 0; waited < 3000; waited++) {          // up to 5 minutes
      const pid_t w = ::waitpid(pid, &status, WNOHANG);
      if (w == pid || (w < 0 && errno != EINTR)) { done = true; break; }
      struct timespec ts = {0, 100 * 1000 * 1000};
      ::nanosleep(&ts, nullptr);
    }
    if (!done) {
      ::kill(pid, SIGKILL);
      (void)::waitpid(pid, &status, 0);
      why = "the compiler process did not finish";
    } else if (WIFEXITED(status) && WEXITSTATUS(status) == 0 && detail::read_file(outp, code) && !code.empty()) {
      result = 1;
    } else {
      std::string log;
      (void)detail::read_file(logp, log);
      why = "compilation failed (" + helper + "): " + log.substr(0, 1500);
    }
  }
  (void)::unlink(in.c_str());
  (void)::unlink(outp.c_str());
  (void)::unlink(logp.c_str());
  return result;
}

// the sequence's translation unit -> a gfx950 code object.  Needs no device (`arch` names the target), which is how
// the CPU test suite checks that the kernel source still compiles this way (mpc_jit_compile_check).  Through the helper
// process; in this process (hiprtc loaded with dlopen) only when there is no helper to start.
inline bool compile(const std::string &source, const std::string &arch, const std::string &src_dir, std::string &code, std::string &why)
{
  const int h = compile_in_helper(source, arch, src_dir, code, why);
  if (std::getenv("MPC_JIT_DEBUG"))
    std::fprintf(stderr, "libmpc_hip: run-time compilation %s\n", h == 1 ? "in the helper process: ok" : h == 0 ? "in the helper process: FAILED" : "in this process (no helper)");
  if (h >= 0) return h == 1;
  detail::Rtc &rtc = detail::rtc();
  if (!rtc.ok()) { why = "neither mpc_jitc (next to the library) nor libhiprtc.so is available"; return false; }
  hiprtcProgram prog = nullptr;
  if (rtc.create(&prog, source.c_str(), "mpc_jit_sequence.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { why = "hiprtcCreateProgram failed"; return false; }
  const std::string o_arch = "--offload-arch=" + arch, o_inc = "-I" + src_dir;
  const char *copts[] = {o_arch.c_str(), "-O3", "-std=c++17", o_inc.c_str()};
  const hiprtcResult r = rtc.compile(prog, 4, copts);
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    (void)rtc.log_size(prog, &n);
    std::string log(n, '\0');
    if (n) (void)rtc.log(prog, &log[0]);
    why = "hiprtc compilation failed: " + log.substr(0, 1500);
    (void)rtc.destroy(&prog);
    return false;
  }
  size_t n = 0;
  (void)rtc.code_size(prog, &n);
  code.resize(n);
  (void)rtc.code(prog, &code[0]);
  (void)rtc.destroy(&prog);
  return true;
}

// Compile (or take from the cache) and load the two kernels of P's sequence on the current device.
// Returns false with a reason in `why`; the caller then runs the run-time module loop.
inline bool build(const MpcVpcParams &P, int testing, Kernels &out, std::string &why, const unsigned char *gtab = nullptr)
{
  unsigned ring_cfg = 0;
  const size_t smem = mpc_vpc_lane_ring_plan(&P, &ring_cfg, nullptr);
  if (smem == 0) { why = "the line rings do not fit the LDS beside the histogram"; return false; }
  const std::string src_dir = source_dir();
  std::string all_sources;
  for (const char *f : {"mpc_vpc_lane.hip", "mpc_kernel_common.h", "mpc_ring.h", "mpc_device.h"}) {
    std::string text;
    if (!detail::read_file(src_dir + "/" + f, text)) { why = "kernel source " + src_dir + "/" + f + " not found"; return false; }
    all_sources += text;
  }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { why = "no device"; return false; }
  const std::string arch = prop.gcnArchName;
  if (arch.compare(0, 6, "gfx950") != 0) { why = "device is " + arch + ", the kernels are written for gfx950"; return false; }
  const std::string source = source_of(P, smem, testing, gtab);

  // ---- cache: keyed by everything the code object depends on ----
  // (the compiler's version enters through the HIP runtime's: they are installed together)
  int rtv = 0;
  (void)hipRuntimeGetVersion(&rtv);
  char key[32];
  std::snprintf(key, sizeof key, "%016llx", detail::fnv1a(source + all_sources + "|" + arch + "|" + std::to_string(rtv)));
  const std::string cdir = detail::cache_dir();
  const std::string cpath = cdir.empty() ? "" : cdir + "/lane_" + key + ".hsaco";
  const std::string mem_key = std::string(key) + "@" + std::to_string(dev);
  std::lock_guard<std::mutex> lock(loaded_mutex());
  {
    auto it = loaded_modules().find(mem_key);
    if (it != loaded_modules().end()) {        // this process has it loaded on this device already
      out.mod = it->second->mod;
      out.stats = it->second->stats;
      out.lines = it->second->lines;
      out.from_cache = true;
      return true;
    }
  }
  auto load = [&](const std::string &code) -> hipError_t {
    Loaded *L = new Loaded;
    L->image = code;                            // the runtime may keep pointing into the image: it lives as long as the module
    hipError_t e = hipModuleLoadData(&L->mod, L->image.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&L->stats, L->mod, "mpc_jit_stats");
    if (e == hipSuccess) e = hipModuleGetFunction(&L->lines, L->mod, "mpc_jit_lines");
    if (e != hipSuccess) {
      if (L->mod) (void)hipModuleUnload(L->mod);
      delete L;
      return e;
    }
    loaded_modules()[mem_key] = L;
    out.mod = L->mod;
    out.stats = L->stats;
    out.lines = L->lines;
    return e;
  };
  std::string code;
  if (!cpath.empty() && detail::read_file(cpath, code) && !code.empty()) {
    if (load(code) == hipSuccess) {
      out.from_cache = true;
      return true;
    }
    (void)std::remove(cpath.c_str());      // a stale or damaged cache entry: compile again
  }
  if (!compile(source, arch, src_dir, code, why)) return false;
  const hipError_t e = load(code);
  if (e != hipSuccess) { why = std::string("loading the compiled kernels failed: ") + hipGetErrorString(e); return false; }
  if (!cpath.empty()) {
    // written under a private name and renamed: another process may be doing the same
    detail::mkdirs(cdir);
    const std::string tmp = cpath + "." + std::to_string((long)getpid());
    std::ofstream f(tmp, std::ios::binary);
    if (f) {
      f.write(code.data(), (std::streamsize)code.size());
      f.close();
      if (!f || std::rename(tmp.c_str(), cpath.c_str()) != 0) (void)std::remove(tmp.c_str());
    }
  }
  return true;
}

}  // namespace mpcjit
