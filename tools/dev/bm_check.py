#!/usr/bin/env python3
"""Development: byte-major configurations on the run-time compiled unrolled kernels -- compile check (CPU) or parity + time (GPU).
    python tools/dev/bm_check.py compile | run [L ...]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mpc = importlib.import_module("cal_22-mpc_amd"); C = importlib.import_module("cal_22-mpc_amd.configs")
traces = importlib.import_module("cal_22-mpc_amd.traces")
mode = sys.argv[1]
az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}
def cases(L):
    def bm(ts=8 * L): return {"TableSize": ts, "Rows": [i % 8 for i in range(ts)], "Cols": [i // 8 for i in range(ts)]}
    prev4 = [max(i - 4, 0) for i in range(L)]; prev1 = [max(i - 1, 0) for i in range(L)]
    w2 = [[1.0, 0.5][i % 2] for i in range(L)]; d1 = [1 if i % 4 == 0 else 0 for i in range(L)]; diff = [(-2 + (i % 5)) for i in range(L)]
    out = []
    for ts in (8 * L, 8 * L - 20, 5 * L + 3, 24, 16):
        s = bm(ts)
        out.append((f"probe modules, byte-major, TableSize {ts}", C.make_config(L, [az, aws, C.one_base(L, 0, True, s), C.consecutive_base(L, 0, True, s),
                    C.diff_base(L, prev4, d1, 0, False, s), C.weight_base(L, prev4, w2, 0, True, s)])))
    s = bm()
    out.append(("OB DF(i-1) WT OB, byte-major", C.make_config(L, [az, C.one_base(L, 0, True, s), C.diff_base(L, prev1, diff, 0, False, s),
                C.weight_base(L, prev4, w2, 0, False, s), C.one_base(L, 0, False, s)])))
    out.append(("single CS, byte-major", C.make_config(L, [az, aws, C.consecutive_base(L, 0, True, s)])))
    out.append(("root 6: run-time loop", C.make_config(L, [az, aws, C.one_base(L, 6, True, s), C.consecutive_base(L, 0, False, s)])))
    return out
bad = 0
for L in [int(a) for a in sys.argv[2:]] or [64]:
    if mode == "compile":
        for name, cfg in cases(L):
            d = mpc.describe_config(cfg)
            print(L, name, d["sequence"], d["compiled"], d["scan_order"], "code", mpc.jit_compile_check(cfg), flush=True)
        continue
    from oracle import oracle as O
    rng = np.random.default_rng(11 + L)
    lines = np.concatenate([traces.structured(6000, L, seed=17), traces.mixed(3000, L), traces.random_u32(1500, L), traces.sine_f32(2048, L),
                            traces.counters_u32(500, L), traces.zeros(70, L), traces.word_same(70, L)])
    lines = lines[rng.permutation(len(lines))]
    for name, cfg in cases(L):
        o = O.VpcOracle(cfg); s_ref, k_ref = o.compress(lines)
        ev = mpc.VPC(cfg); s, k = ev.compress_lines(lines)
        ok = bool((s == s_ref).all() and (k == k_ref).all() and (ev.stats_vector() == o.stats_vector()).all())
        ev.reset(); ev.compress_lines(lines, want_sizes=False, want_selected=False)
        ok = ok and bool((ev.stats_vector() == o.stats_vector()).all())
        mism = np.nonzero((s != s_ref) | (k != k_ref))[0]
        print(f"L={L} {name:44s} {ev.kernel_form:34s} parity {'ok' if ok else 'FAIL'}" + ("" if ok else f" {len(mism)} lines, first {mism[:3]}: {s[mism[:3]]} vs {s_ref[mism[:3]]}, sel {k[mism[:3]]} vs {k_ref[mism[:3]]}"), flush=True)
        bad += 0 if ok else 1
        ev.close()
    if mode == "run" and L == 64:
        import torch
        n = (4 << 30) // L
        st = torch.cuda.Stream()
        for wl in ("random_u32", "mixed"):
            buf = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
            mpc.synth_fill(buf.data_ptr(), n, L, wl); torch.cuda.synchronize()
            for jit in ("1", "0"):
                os.environ["MPC_JIT"] = jit
                name, cfg = cases(L)[0]
                ev = mpc.VPC(cfg)
                for _ in range(3): ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                for _ in range(5): ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
                b.record(st); torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 5
                print(f"{wl:11s} {name} [{ev.kernel_form}] {n * L / ms / 1e6 / 8000:.3f} of peak", flush=True)
                ev.close()
            os.environ.pop("MPC_JIT", None)
print("FAILED" if bad else "all ok")
if bad: raise SystemExit(1)
