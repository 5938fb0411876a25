#!/usr/bin/env python3
"""Development helper: kernel time of VPC configurations that take the fast kernel's run-time module loop
(non-zero RootIndex, truncated scan tables, the per-datatype model set) and the generic kernel, 64-byte lines."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpc = importlib.import_module("cal_22-mpc_amd"); C = importlib.import_module("cal_22-mpc_amd.configs")
L = 64
n = (4 << 30) // L
def trunc(ts): return {"TableSize": ts, "Rows": [i // L for i in range(ts)], "Cols": [i % L for i in range(ts)]}
prev4 = [max(i - 4, 0) for i in range(L)]; w2 = [[1.0, 0.5][i % 2] for i in range(L)]; d1 = [1 if i % 4 == 0 else 0 for i in range(L)]
az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}
perm = [int(x) for x in __import__("numpy").random.default_rng(1).permutation(8 * L)]
cfgs = {
    "probe (unrolled)": C.probe_config(L),
    "mpc_config (5 models)": C.mpc_config(L),
    "probe, roots 5/0/3/2": C.make_config(L, [az, aws, C.one_base(L, 5, True), C.consecutive_base(L, 0, True), C.diff_base(L, prev4, d1, 3, False), C.weight_base(L, prev4, w2, 2, True)]),
    "probe, roots 40/0/17/33": C.make_config(L, [az, aws, C.one_base(L, 40, True), C.consecutive_base(L, 0, True), C.diff_base(L, prev4, d1, 17, False), C.weight_base(L, prev4, w2, 33, True)]),
    "probe, TableSize 6L": C.make_config(L, [az, aws, C.one_base(L, 0, True, trunc(6 * L)), C.consecutive_base(L, 0, True, trunc(6 * L)), C.diff_base(L, prev4, d1, 0, False, trunc(6 * L)), C.weight_base(L, prev4, w2, 0, True, trunc(6 * L))]),
    "int32 model alone": C.datatype_config(L, "int32"),
    "probe, byte-major scan": C.make_config(L, [az, aws] + [f(*a, {"TableSize": 8 * L, "Rows": [i % 8 for i in range(8 * L)], "Cols": [i // 8 for i in range(8 * L)]})
                                                          for f, a in ((C.one_base, (L, 0, True)), (C.consecutive_base, (L, 0, True)), (C.diff_base, (L, prev4, d1, 0, False)), (C.weight_base, (L, prev4, w2, 0, True)))]),
    "probe, permuted scan (generic)": C.make_config(L, [az, aws, C.one_base(L, 0, True, {"TableSize": 8 * L, "Rows": [p // L for p in perm], "Cols": [p % L for p in perm]}), C.consecutive_base(L, 0, True), C.diff_base(L, prev4, d1, 0, False), C.weight_base(L, prev4, w2, 0, True)]),
}
st = torch.cuda.Stream()
for wl in ("random_u32", "mixed"):
    buf = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
    mpc.synth_fill(buf.data_ptr(), n, L, wl); torch.cuda.synchronize()
    for name, cfg in cfgs.items():
        ev = mpc.VPC(cfg)
        m = n if ev.kernel_path == mpc.MPC_PATH_VPC_FAST else n // 64
        ev.compress_device(buf.data_ptr(), m, stream=st.cuda_stream); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(3): ev.compress_device(buf.data_ptr(), m, stream=st.cuda_stream)
        b.record(st); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        d = mpc.describe_config(cfg)
        print(f"{wl:11s} {name:32s} path {d['path']:7s} {d['sequence']:13s} {d.get('compiled', ''):11s} {m * L / ms / 1e6:8.0f} GB/s  {m * L / ms / 1e6 / 8000:.3f} of peak")
        ev.close()
