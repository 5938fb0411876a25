#!/usr/bin/env python3
"""Development: one VPC configuration of tools/time_layouts.py's set, launched K times on 16 GiB (for rocprofv3 --pmc).
    python3 tools/dev/rt_run.py {roots|trunc|bm|probe} [workload] [K]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mpc = importlib.import_module("cal_22-mpc_amd"); C = importlib.import_module("cal_22-mpc_amd.configs")
which = sys.argv[1]; wl = sys.argv[2] if len(sys.argv) > 2 else "random_u32"; K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
L = 64; n = (16 << 30) // L
def trunc(ts): return {"TableSize": ts, "Rows": [i // L for i in range(ts)], "Cols": [i % L for i in range(ts)]}
prev4 = [max(i - 4, 0) for i in range(L)]; w2 = [[1.0, 0.5][i % 2] for i in range(L)]; d1 = [1 if i % 4 == 0 else 0 for i in range(L)]
az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}
bm = {"TableSize": 8 * L, "Rows": [i % 8 for i in range(8 * L)], "Cols": [i // 8 for i in range(8 * L)]}
cfg = {"probe": lambda: C.probe_config(L),
       "roots": lambda: C.make_config(L, [az, aws, C.one_base(L, 5, True), C.consecutive_base(L, 0, True), C.diff_base(L, prev4, d1, 3, False), C.weight_base(L, prev4, w2, 2, True)]),
       "trunc": lambda: C.make_config(L, [az, aws, C.one_base(L, 0, True, trunc(6 * L)), C.consecutive_base(L, 0, True, trunc(6 * L)), C.diff_base(L, prev4, d1, 0, False, trunc(6 * L)), C.weight_base(L, prev4, w2, 0, True, trunc(6 * L))]),
       "bm": lambda: C.make_config(L, [az, aws, C.one_base(L, 0, True, bm), C.consecutive_base(L, 0, True, bm), C.diff_base(L, prev4, d1, 0, False, bm), C.weight_base(L, prev4, w2, 0, True, bm)])}[which]()
st = torch.cuda.Stream()
buf = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
mpc.synth_fill(buf.data_ptr(), n, L, wl); torch.cuda.synchronize()
ev = mpc.VPC(cfg)
for _ in range(2): ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(st)
for _ in range(K): ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
b.record(st); torch.cuda.synchronize()
ms = a.elapsed_time(b) / K
print(f"{which} {wl}: {ms:.3f} ms per 16 GiB, {n * L / ms / 1e6 / 8000:.3f} of peak, ratio {ev.ratio() if hasattr(ev, 'ratio') else ''}")
ev.close()
