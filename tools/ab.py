#!/usr/bin/env python3
"""Development only: same-box A/B timing of variant builds of libmpc_hip.so.

    tools/ab.py build NAME[@CSRC_DIR][=FLAGS] ...       # here: tools/ablate/libmpc_hip_NAME.so, each with -DMPC_DEV_ONLY64 FLAGS
    tools/ab.py run [--rounds R] [--workloads a,b] [--algo VPC] NAME ...   # on the GPU box (one gpurun call)

`run` starts one child process per (variant, round) -- a process binds one library -- which first checks
the variant against the CPU oracle on a mixed bag of 64-byte lines, then times the kernel on each
workload (256 Mi lines resident, HIP events, like bench.py).  Variants are interleaved across rounds.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "ablate")
CSRC = os.path.join(ROOT, "cal_22-mpc_amd", "csrc")


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition("=")
        name, _, src = name.partition("@")          # NAME@DIR: sources from another csrc directory (a saved baseline)
        csrc = os.path.abspath(src) if src else CSRC
        # NAME: the timed library; NAME_t: the same with -DMPC_TESTING=1 (honours MPC_TEST_GRID) for the capped-grid parity check
        for suffix, extra in (("", []), ("_t", ["-DMPC_TESTING=1"])):
            so = os.path.join(OUT, f"libmpc_hip_{name}{suffix}.so")
            cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DMPC_DEV_ONLY64", *flags.split(), *extra,
                   "-shared", "-o", so] + [os.path.join(csrc, f) for f in ("mpc_vpc_lane.hip", "mpc_kernels.hip", "mpc_capi.hip")]
            procs.append((name + suffix, subprocess.Popen(cmd)))
    bad = [n for n, p in procs if p.wait() != 0]
    if bad:
        raise SystemExit(f"build failed: {bad}")
    print("built", [n for n, _ in procs])


CHILD = r'''
import importlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["MPC_ROOT"])
mpc = importlib.import_module("cal_22-mpc_amd")
configs = importlib.import_module("cal_22-mpc_amd.configs")
traces = importlib.import_module("cal_22-mpc_amd.traces")
algo = os.environ["AB_ALGO"]; workloads = os.environ["AB_WORKLOADS"].split(","); n = int(os.environ["AB_LINES"])
res = {"variant": os.environ["AB_NAME"]}
def make(L):
    return mpc.VPC(configs.probe_config(L)) if algo == "VPC" else getattr(mpc, algo)(L)
if os.environ.get("AB_CHECK") == "1":
    from oracle import oracle as O
    lines = np.concatenate([traces.zeros(70), traces.word_same(70), traces.random_u32(3000), traces.sine_f32(2048),
                            traces.mixed(3000), traces.structured(6000), traces.bdi_stress(1400), traces.pointers_u64(500, 64)])
    lines = lines[np.random.default_rng(5).permutation(len(lines))]
    ev = make(64)
    if algo == "VPC":
        o = O.VpcOracle(configs.probe_config(64)); s_ref, k_ref = o.compress(lines)
        s, k = ev.compress_lines(lines)
        ok = bool((s == s_ref).all() and (k == k_ref).all() and (ev.stats_vector() == o.stats_vector()).all())
        # lines of two kinds alternating (paired groups), in trace order, with a ragged end
        alt = np.concatenate([traces.mixed(20000), traces.random_u32(777), traces.mixed(5001, first_line=3)])
        ev.reset(); o.reset()
        s_ref, k_ref = o.compress(alt)
        s, k = ev.compress_lines(alt)
        ok = ok and bool((s == s_ref).all() and (k == k_ref).all() and (ev.stats_vector() == o.stats_vector()).all())
    else:
        o = {"BDI": O.BdiOracle, "FPC": O.FpcOracle, "BPC": O.BpcOracle}[algo](64)
        r = o.compress(lines); s_ref = r[0] if isinstance(r, tuple) else r
        s, _ = ev.compress_lines(lines)
        ok = bool((s == s_ref).all() and (ev.stats_vector() == o.stats_vector()).all())
    res["parity"] = ok
    ev.close()
W = {"random_u32": ("random_u32", 64), "sine_f32": ("sine_f32", 64), "mixed": ("mixed", 64), "zeros": ("zeros", 64),
     "pointers_u64_128": ("pointers_u64", 128), "random_u32_32": ("random_u32", 32), "mixed_32": ("mixed", 32)}
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
for w in [x for x in workloads if x]:
    kind, L = W[w]
    n = int(os.environ["AB_LINES"]) * 64 // L
    buf = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mpc.synth_fill(buf.data_ptr(), n, L, kind, first_line=0)
    torch.cuda.synchronize()
    ev = make(L)
    for _ in range(2):
        ev.compress_device(buf.data_ptr(), n, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    ms = []
    for _ in range(int(os.environ["AB_STEPS"])):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); ev.compress_device(buf.data_ptr(), n, stream=stream.cuda_stream); b.record(stream)
        torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    v = ev.stats_vector()
    res[w] = {"avg": sum(ms) / len(ms), "min": min(ms), "ratio": float(v[1]) / float(v[2])}
    ev.close(); del buf
print("AB " + json.dumps(res), flush=True)
'''


def run(args):
    rounds, workloads, algo, lines, steps = 2, "random_u32,sine_f32,mixed", "VPC", 256 << 20, 8
    nocheck = False
    names = []
    it = iter(args)
    for a in it:
        if a == "--rounds": rounds = int(next(it))
        elif a == "--workloads": workloads = next(it)
        elif a == "--algo": algo = next(it)
        elif a == "--lines": lines = int(next(it))
        elif a == "--steps": steps = int(next(it))
        elif a == "--no-check": nocheck = True          # timing ablations (results are wrong by design)
        else: names.append(a)
    table = {}
    for r in range(rounds):
        for name in names:
            so = os.path.join(OUT, f"libmpc_hip_{name}.so") if name != "tree" else os.path.join(ROOT, "cal_22-mpc_amd", "libmpc_hip.so")
            env = dict(os.environ, MPC_HIP_LIB=so, MPC_ROOT=ROOT, AB_NAME=name, AB_ALGO=algo, AB_WORKLOADS=workloads,
                       AB_LINES=str(lines), AB_STEPS=str(steps), AB_CHECK="1" if (r == 0 and not nocheck) else "0")
            if r == 0 and not nocheck:
                # parity once more with the grid capped to 3 workgroups (a wave then walks many groups of lines: ring
                # stages are reused, queues fill and drain), in a process of its own: the cap is read once per process
                so_t = so[:-3] + "_t.so"
                cenv = dict(env, AB_WORKLOADS="", MPC_TEST_GRID="3", MPC_HIP_LIB=so_t if os.path.exists(so_t) else so)
                cp = subprocess.run([sys.executable, "-c", CHILD], env=cenv, capture_output=True, text=True)
                cl = [l for l in cp.stdout.split("\n") if l.startswith("AB ")]
                if not cl or not json.loads(cl[-1][3:]).get("parity"):
                    print(f"{name}: parity with a capped grid FAILED\n{cp.stdout[-1500:]}\n{cp.stderr[-1500:]}", flush=True)
                    raise SystemExit(3)
            p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.split("\n") if l.startswith("AB ")]
            if not line:
                print(f"{name} round {r}: FAILED\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}", flush=True)
                raise SystemExit(3)          # a variant that crashed may have faulted on the GPU: no further GPU step in this call
            d = json.loads(line[-1][3:])
            table.setdefault(name, []).append(d)
            print(f"round {r} {name:>12}: " + ("" if "parity" not in d else f"parity={'ok' if d['parity'] else 'FAIL'} ") +
                  " ".join(f"{w}={d[w]['avg']:.3f}/{d[w]['min']:.3f}" for w in workloads.split(",")), flush=True)
    print("---- mean of avg ms over rounds")
    for name, rs in table.items():
        print(f"{name:>12}: " + " ".join(f"{w}={sum(r[w]['avg'] for r in rs) / len(rs):.3f}" for w in workloads.split(",")))


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in ("build", "run"):
        raise SystemExit(__doc__)
    (build if sys.argv[1] == "build" else run)(sys.argv[2:])
