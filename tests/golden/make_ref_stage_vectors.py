#!/usr/bin/env python3
"""Generate tests/golden/ref_stage_vectors.json from the REFERENCE's own stage
classes (oracle/_ref/libmpc_refstages.so = PredictorModule.cpp,
ResidueModule.cpp, XORModule.cpp, ScanModule.cpp, FPCModule.cpp compiled
unmodified from /root/reference by oracle/Makefile).

Run in the build container only (needs /root/reference):
    make -C oracle _ref && python tests/golden/make_ref_stage_vectors.py

The output holds inputs and the reference's outputs only (no reference text).
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def module_specs(L, rng):
    ident_rows = [i // L for i in range(8 * L)]
    ident_cols = [i % L for i in range(8 * L)]
    perm = rng.permutation(8 * L)
    specs = []
    # kind, root, base, weight, diff, consecutive_xor, table
    specs.append(dict(kind=2, root=0, cx=1, rows=ident_rows, cols=ident_cols))
    specs.append(dict(kind=2, root=5, cx=0, rows=ident_rows, cols=ident_cols))
    specs.append(dict(kind=3, root=0, cx=1, rows=ident_rows, cols=ident_cols))
    base4 = [max(i - 4, 0) for i in range(L)]
    specs.append(dict(kind=1, root=0, base=base4, diff=[1 if i % 4 == 0 else 0 for i in range(L)],
                      cx=0, rows=ident_rows, cols=ident_cols))
    specs.append(dict(kind=0, root=0, base=base4, weight=[1.0 if i % 2 == 0 else 0.5 for i in range(L)],
                      cx=1, rows=ident_rows, cols=ident_cols))
    # arbitrary tables, non-zero root, permuted (and truncated) scan table
    rb = [int(x) for x in rng.integers(0, L, L)]
    rd = [int(x) for x in rng.integers(-300, 300, L)]
    specs.append(dict(kind=1, root=3, base=rb, diff=rd, cx=1,
                      rows=[int(p) // L for p in perm], cols=[int(p) % L for p in perm]))
    rw = [float(2.0 ** int(x)) for x in rng.integers(-9, 10, L)]
    specs.append(dict(kind=0, root=L - 1, base=rb, weight=rw, cx=0,
                      rows=[int(p) // L for p in perm[: 8 * L - 24]],
                      cols=[int(p) % L for p in perm[: 8 * L - 24]]))
    return specs


# WeightBase tables whose weights are NOT powers of two, and extreme ones: the reference takes
# (int)log2f(w), truncating toward zero (PredictorModule.cpp:30) -- 0.3 -> -1, 0.7 -> 0, 3.0 -> 1,
# 5.5 -> 2, 300.0 -> 8 (the byte shifts out), 0.001 -> -9 (likewise)
ODD_WEIGHTS = [0.3, 0.7, 3.0, 5.5, 300.0, 0.001, 1.0, 0.5, 1.5, 0.99, 1.99, 2.0, 4.01, 0.26, 0.24, 127.9,
               128.0, 0.0079, 0.0078125, 255.0, 0.126]


def weight_specs(L, rng):
    """Appended after module_specs() with a generator of its own, so that the earlier vectors
    stay byte-identical when this file is regenerated."""
    ident_rows = [i // L for i in range(8 * L)]
    ident_cols = [i % L for i in range(8 * L)]
    base4 = [max(i - 4, 0) for i in range(L)]
    specs = []
    # windowed tables (previous word), two shift distances: the lane kernel's WeightBase forms
    for pair in ((0.3, 0.7), (3.0, 1.0), (5.5, 0.7), (300.0, 1.99), (0.001, 0.99), (0.26, 0.24)):
        specs.append(dict(kind=0, root=0, base=base4, weight=[pair[i % 2] for i in range(L)], cx=1,
                          rows=ident_rows, cols=ident_cols))
    # every listed weight somewhere in one table, arbitrary base bytes, both XOR flavours, roots 0 and 7
    for root, cx in ((0, 0), (7, 1)):
        rb = [int(x) for x in rng.integers(0, L, L)]
        rw = [float(ODD_WEIGHTS[int(x)]) for x in rng.integers(0, len(ODD_WEIGHTS), L)]
        specs.append(dict(kind=0, root=root, base=rb, weight=rw, cx=cx, rows=ident_rows, cols=ident_cols))
    return specs


def lines_for(L, rng):
    out = []
    out.append(rng.integers(0, 256, L, dtype=np.uint8))
    out.append(rng.integers(0, 256, L, dtype=np.uint8))
    t = np.arange(L // 4)
    out.append(np.sin(2 * np.pi * (t + 37) / 1024).astype("<f4").view(np.uint8))
    out.append(((t * 3 + 100) % 1000).astype("<u4").view(np.uint8))
    small = np.zeros(L, dtype=np.uint8)
    small[::4] = rng.integers(0, 4, L // 4)
    out.append(small)
    one = np.zeros(L, dtype=np.uint8)
    one[7] = 0x80
    out.append(one)
    out.append(np.full(L, 0xA5, dtype=np.uint8))
    ramp = (np.arange(L) * 2 + 9).astype(np.uint8)
    out.append(ramp)
    return out


def main():
    R = O.ref_lib()
    if R is None:
        raise SystemExit("oracle/_ref is not built (make -C oracle _ref)")
    rng = np.random.default_rng(20221)
    rng_w = np.random.default_rng(20222)
    vectors = []
    jobs = []
    for L in (32, 64, 128):
        specs = module_specs(L, rng)
        lines = lines_for(L, rng)
        jobs.append((L, specs, lines))
    for L, _, lines in list(jobs):
        jobs.append((L, weight_specs(L, rng_w), lines))
    for L, specs, lines in jobs:
        for s in specs:
            base = np.array(s.get("base", [0] * L), dtype=np.int32)
            weight = np.array(s.get("weight", [1.0] * L), dtype=np.float32)
            diff = np.array(s.get("diff", [0] * L), dtype=np.int32)
            rows = np.array(s["rows"], dtype=np.int32)
            cols = np.array(s["cols"], dtype=np.int32)
            for ln in lines:
                ln = np.ascontiguousarray(ln, dtype=np.uint8)
                pred = np.zeros(L, dtype=np.uint8)
                res = np.zeros(L, dtype=np.uint8)
                sc = np.zeros(8 * L // 16, dtype=np.uint16)
                mae, mse = C.c_double(), C.c_double()
                a = (s["kind"], s["root"], L, base.ctypes.data, weight.ctypes.data, diff.ctypes.data)
                R.ref_predict(*a, ln.ctypes.data, pred.ctypes.data)
                R.ref_residue(*a, ln.ctypes.data, res.ctypes.data)
                R.ref_mae_mse(*a, ln.ctypes.data, C.byref(mae), C.byref(mse))
                R.ref_scanned(*a, s["cx"], len(rows), rows.ctypes.data, cols.ctypes.data,
                              ln.ctypes.data, sc.ctypes.data)
                fpc = R.ref_fpc_size(sc.ctypes.data, len(sc))
                vectors.append(dict(
                    L=L, kind=s["kind"], root=s["root"], cx=s["cx"],
                    base=base.tolist() if "base" in s else None,
                    weight=weight.tolist() if "weight" in s else None,
                    diff=diff.tolist() if "diff" in s else None,
                    scan=None if (s["rows"] == [i // L for i in range(8 * L)] and
                                  s["cols"] == [i % L for i in range(8 * L)]) else
                    dict(rows=rows.tolist(), cols=cols.tolist()),
                    line=ln.tobytes().hex(), pred=pred.tobytes().hex(), residue=res.tobytes().hex(),
                    scanned=[int(x) for x in sc], fpc=int(fpc),
                    mae=float(mae.value).hex(), mse=float(mse.value).hex()))
    # common-encoder vectors: hand-picked row patterns (FPCModule.cpp:19-85)
    enc = []
    pats = [0x0000, 0x8000, 0x0001, 0x0180, 0x0003, 0xC000, 0x0101, 0x00FF, 0xFF00, 0x8001,
            0x0005, 0x0300, 0x0081, 0x7FFF, 0x0100, 0x0080]
    for trial in range(200):
        rows = int(rng.choice([16, 32, 64]))
        v = rng.choice(pats, rows).astype(np.uint16)
        if trial % 3 == 0:
            v[rng.integers(0, rows, rows // 2)] = 0
        if trial % 5 == 0:
            v = rng.integers(0, 1 << 16, rows).astype(np.uint16)
        v = np.ascontiguousarray(v)
        enc.append(dict(rows=[int(x) for x in v], size=int(R.ref_fpc_size(v.ctypes.data, rows))))
    out = dict(
        provenance="outputs of the reference's own PredictorModule/ResidueModule/XORModule/"
                   "ScanModule/FPCModule classes (compiled unmodified, g++ -O3) driven by "
                   "oracle/ref_stage_harness.cpp; generated by tests/golden/make_ref_stage_vectors.py",
        stage_vectors=vectors, encoder_vectors=enc)
    path = os.path.join(ROOT, "tests", "golden", "ref_stage_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, len(vectors), "stage vectors,", len(enc), "encoder vectors,",
          os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
