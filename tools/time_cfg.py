#!/usr/bin/env python3
"""Development helper: time one named configuration (16 GiB, 64 B lines): tools/time_cfg.py NAME WORKLOAD"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpc = importlib.import_module("cal_22-mpc_amd"); C = importlib.import_module("cal_22-mpc_amd.configs")
L = 64; n = (16 << 30) // L
name, wl = sys.argv[1], sys.argv[2]
prev4 = [max(i - 4, 0) for i in range(L)]; w2 = [[1.0, 0.5][i % 2] for i in range(L)]; d1 = [1 if i % 4 == 0 else 0 for i in range(L)]
az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}
cfg = {"probe": lambda: C.probe_config(L), "mpc": lambda: C.mpc_config(L), "int32": lambda: C.datatype_config(L, "int32"),
       "roots": lambda: C.make_config(L, [az, aws, C.one_base(L, 5, True), C.consecutive_base(L, 0, True), C.diff_base(L, prev4, d1, 3, False), C.weight_base(L, prev4, w2, 2, True)])}[name]()
buf = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
mpc.synth_fill(buf.data_ptr(), n, L, wl); torch.cuda.synchronize()
ev = mpc.VPC(cfg); st = torch.cuda.Stream()
for _ in range(2): ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(st)
for _ in range(4): ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
b.record(st); torch.cuda.synchronize()
print(name, wl, mpc.describe_config(cfg)["sequence"], f"{a.elapsed_time(b) / 4:.3f} ms")
