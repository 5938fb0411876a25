#!/usr/bin/env python3
"""Development: parity of the general-layout twins against the oracle, and their kernel time (GPU box).
    MPC_HIP_LIB=tools/ablate/libmpc_hip_gen.so python tools/dev/gen_check.py [L ...]"""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mpc = importlib.import_module("cal_22-mpc_amd"); C = importlib.import_module("cal_22-mpc_amd.configs")
traces = importlib.import_module("cal_22-mpc_amd.traces")
from oracle import oracle as O
az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}
bad = 0
for L in [int(a) for a in sys.argv[1:]] or [64]:
    def trunc(ts): return None if ts is None else {"TableSize": ts, "Rows": [i // L for i in range(ts)], "Cols": [i % L for i in range(ts)]}
    prev4 = [max(i - 4, 0) for i in range(L)]; prev8 = [max(i - 8, 0) for i in range(L)]
    w2 = [[1.0, 0.5][i % 2] for i in range(L)]; d1 = [1 if i % 4 == 0 else 0 for i in range(L)]
    rng = np.random.default_rng(7 + L)
    lines = np.concatenate([traces.structured(6000, L, seed=17), traces.mixed(3000, L), traces.random_u32(1000, L), traces.sine_f32(2048, L),
                            traces.counters_u32(500, L), traces.zeros(70, L), traces.word_same(70, L)])
    lines = lines[rng.permutation(len(lines))]
    cases = []
    for roots, ts in (((5, 0, 3, 2), None), ((0, 0, 0, 0), 6 * L), ((15, 0, 1, 3), 7 * L), ((1, 0, 0, 0), L), ((12, 0, 2, 1), 3 * L),
                      ((4, 0, 3, 0), None), ((8, 0, 0, 2), 2 * L), ((0, 0, 3, 3), None), ((16, 0, 0, 0), None), ((0, 0, 0, 0), 6 * L + 8)):
        s = trunc(ts)
        mods = [az, aws, C.one_base(L, roots[0], True, s), C.consecutive_base(L, 0, True, s), C.diff_base(L, prev4, d1, roots[2], False, s),
                C.weight_base(L, prev4, w2, roots[3], True, s)]
        cases.append((f"probe roots {roots} TableSize {ts}", C.make_config(L, mods)))
    for roots, ts in (((7, 0, 5, 6), 5 * L), ((3, 0, 0, 0), None)):
        s = trunc(ts)
        mods = [az, aws, C.one_base(L, roots[0], True, s), C.consecutive_base(L, 0, True, s), C.diff_base(L, prev8, d1, roots[2], False, s),
                C.weight_base(L, prev8, w2, roots[3], True, s)]
        cases.append((f"i-8 tables roots {roots} TableSize {ts}", C.make_config(L, mods)))
    for name, cfg in cases:
        d = mpc.describe_config(cfg)
        o = O.VpcOracle(cfg); s_ref, k_ref = o.compress(lines)
        ev = mpc.VPC(cfg); s, k = ev.compress_lines(lines)
        ok = bool((s == s_ref).all() and (k == k_ref).all() and (ev.stats_vector() == o.stats_vector()).all())
        # the statistics-only kernels too (no per-line outputs)
        ev.reset(); ev.compress_lines(lines, want_sizes=False, want_selected=False)
        ok = ok and bool((ev.stats_vector() == o.stats_vector()).all())
        print(f"L={L} {name:46s} {d['path']:7s} {d['sequence']:13s} general={d.get('general_layout')}  parity {'ok' if ok else 'FAIL'}"
              + ("" if ok else f"  first mismatch at {int(np.argmax((s != s_ref) | (k != k_ref)))}: {s[np.argmax((s != s_ref) | (k != k_ref))]} vs {s_ref[np.argmax((s != s_ref) | (k != k_ref))]}"), flush=True)
        bad += 0 if ok else 1
        ev.close()
print("FAILED" if bad else "all ok")
if bad: raise SystemExit(1)
