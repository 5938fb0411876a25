#!/usr/bin/env python3
"""Development: module sequences of growing length, compiled at creation against the run-time loop (GPU box)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mpc = importlib.import_module("cal_22-mpc_amd"); C = importlib.import_module("cal_22-mpc_amd.configs")
L = 64; n = (4 << 30) // L
az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}
prev1 = [max(i - 1, 0) for i in range(L)]; prev4 = [max(i - 4, 0) for i in range(L)]; prev8 = [max(i - 8, 0) for i in range(L)]
w2 = [[1.0, 0.5][i % 2] for i in range(L)]; diff = [(-2 + (i % 5)) for i in range(L)]
os.environ["MPC_JIT_MAX_MODULES"] = "16"
pool = 2 * [C.diff_base(L, prev8, [1] * L, 0, True), C.one_base(L, 0, False), C.consecutive_base(L, 0, False), C.weight_base(L, prev8, w2, 0, True),
        C.diff_base(L, prev4, diff, 0, False), C.weight_base(L, prev1, w2, 0, False), C.one_base(L, 0, True), C.consecutive_base(L, 0, True)]
st = torch.cuda.Stream()
os.environ["MPC_JIT_CACHE"] = ""
for wl in ("random_u32", "mixed"):
    buf = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
    mpc.synth_fill(buf.data_ptr(), n, L, wl); torch.cuda.synchronize()
    for k in (3, 5, 8, 10, 12, 14):
        cfg = C.make_config(L, [az, aws] + pool[:k])
        out = []
        for jit in ("1", "0"):
            os.environ["MPC_JIT"] = jit
            ev = mpc.VPC(cfg)
            for _ in range(3): ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            for _ in range(4): ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
            b.record(st); torch.cuda.synchronize()
            out.append((ev.kernel_form, n * L / (a.elapsed_time(b) / 4) / 1e6 / 8000))
            ev.close()
        print(f"{wl:11s} {k} modules: {out[0][0]} {out[0][1]:.3f}   {out[1][0]} {out[1][1]:.3f}", flush=True)
