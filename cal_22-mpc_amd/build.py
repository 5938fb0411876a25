"""In-tree build of the native code (gfx950 only).

    python "cal_22-mpc_amd/build.py"          # libmpc_hip.so + bin/compressor
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(HERE, "libmpc_hip.so")
TEST_LIB = os.path.join(HERE, "libmpc_hip_test.so")      # -DMPC_TESTING=1: route counters + MPC_TEST_GRID (tests only)
CLI = os.path.join(ROOT, "bin", "compressor")
JITC = os.path.join(HERE, "mpc_jitc")                   # the run-time compiler's helper process (csrc/mpc_jitc.cpp)

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("MPC_EXTRA_FLAGS", "").split()      # development: extra -D switches for a variant build


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _sources(d: str):
    out = []
    for dirpath, _, files in os.walk(d):
        for f in files:
            if f.endswith((".hip", ".h", ".cpp", ".hpp")):
                out.append(os.path.join(dirpath, f))
    return out


def build_lib(force: bool = False, verbose: bool = False, test: bool = False) -> str:
    """libmpc_hip.so (test=True: libmpc_hip_test.so, the same sources with -DMPC_TESTING=1).  The translation units
    are compiled in parallel: the lane kernel file once per line size (-DMPC_LANE_W=8/16/32) plus its dispatcher
    (-DMPC_LANE_W=0), the other kernels, the C ABI; then linked."""
    LIB = TEST_LIB if test else globals()["LIB"]
    deps = _sources(CSRC) + [os.path.join(ROOT, "include", "mpc_hip.h")]
    if not force and _newer(LIB, deps):
        return LIB
    objdir = os.path.join(HERE, "obj_test" if test else "obj")
    os.makedirs(objdir, exist_ok=True)
    lane = os.path.join(CSRC, "mpc_vpc_lane.hip")
    units = [(lane, f"lane_w{w}.o", [f"-DMPC_LANE_W={w}"]) for w in (16, 32, 8, 0)]
    units += [(os.path.join(CSRC, "mpc_kernels.hip"), "kernels.o", []), (os.path.join(CSRC, "mpc_capi.hip"), "capi.o", [])]
    tflag = ["-DMPC_TESTING=1"] if test else []
    procs = []
    for src, obj, extra in units:
        cmd = [HIPCC, *FLAGS, *tflag, *extra, "-c", src, "-o", os.path.join(objdir, obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + [os.path.join(objdir, obj) for _, obj, _ in units]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


def build_jitc(force: bool = False, verbose: bool = False) -> str:
    """mpc_jitc: hiprtc in a process of its own (csrc/mpc_jit.h starts it when a module sequence has no built-in kernel)."""
    src = os.path.join(CSRC, "mpc_jitc.cpp")
    if not force and _newer(JITC, [src]):
        return JITC
    rocm = os.path.dirname(os.path.dirname(HIPCC)) if os.path.sep in HIPCC else "/opt/rocm"
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"), src,
           "-L", os.path.join(rocm, "lib"), "-lhiprtc", f"-Wl,-rpath,{os.path.join(rocm, 'lib')}", "-o", JITC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return JITC


def build_cli(force: bool = False, verbose: bool = False) -> str:
    """bin/compressor: the reference's CLI (src/main.cpp) over libmpc_hip.so."""
    if not os.path.isdir(HOST):
        return ""
    deps = _sources(HOST) + [LIB]
    if not force and _newer(CLI, deps):
        return CLI
    srcs = [os.path.join(HOST, f) for f in sorted(os.listdir(HOST)) if f.endswith(".cpp")]
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cmd = [HIPCC, "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", CLI, *srcs,
           "-L", HERE, "-lmpc_hip", "-Wl,-rpath,$ORIGIN/../cal_22-mpc_amd"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return CLI


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_lib(force, verbose)
    build_lib(force, verbose, test=True)
    build_jitc(force, verbose)
    build_cli(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
