// mpc_jitc -- the run-time compiler's own process (csrc/mpc_jit.h).
//
//   mpc_jitc SOURCE.hip ARCH INCLUDE_DIR OUT.hsaco
//
// Compiles one translation unit with hiprtc and writes the code object.  libmpc_hip.so starts this program (posix_spawn)
// instead of calling hiprtc itself: the compiler -- a whole LLVM -- then lives in a process of its own, cannot clash with
// another copy of the ROCm libraries the host application may carry (a PyTorch wheel bundles its own libhiprtc /
// libamd_comgr), and a fatal compiler error ends this process only (the library then runs the run-time module loop).
// Exit code 0: OUT written; otherwise the log is on stderr.
#include <hip/hiprtc.h>

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>

int main(int argc, char **argv)
{
  if (argc != 5) {
    std::fprintf(stderr, "usage: mpc_jitc SOURCE.hip ARCH INCLUDE_DIR OUT.hsaco\n");
    return 2;
  }
  std::ifstream f(argv[1], std::ios::binary);
  if (!f) { std::fprintf(stderr, "mpc_jitc: cannot read %s\n", argv[1]); return 2; }
  std::ostringstream ss;
  ss << f.rdbuf();
  const std::string source = ss.str();
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, source.c_str(), "mpc_jit_sequence.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    std::fprintf(stderr, "mpc_jitc: hiprtcCreateProgram failed\n");
    return 1;
  }
  const std::string o_arch = std::string("--offload-arch=") + argv[2], o_inc = std::string("-I") + argv[3];
  const char *opts[] = {o_arch.c_str(), "-O3", "-std=c++17", o_inc.c_str()};
  const hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    (void)hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) (void)hiprtcGetProgramLog(prog, &log[0]);
    std::fprintf(stderr, "mpc_jitc: %s\n%s\n", hiprtcGetErrorString(r), log.c_str());
    return 1;
  }
  size_t n = 0;
  (void)hiprtcGetCodeSize(prog, &n);
  std::string code(n, '\0');
  (void)hiprtcGetCode(prog, &code[0]);
  (void)hiprtcDestroyProgram(&prog);
  std::ofstream out(argv[4], std::ios::binary);
  out.write(code.data(), (std::streamsize)code.size());
  out.close();
  if (!out) { std::fprintf(stderr, "mpc_jitc: cannot write %s\n", argv[4]); return 2; }
  return 0;
}
