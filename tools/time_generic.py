#!/usr/bin/env python3
"""Development helper: throughput of the generic VPC kernel (configuration outside the fast path)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpc = importlib.import_module("cal_22-mpc_amd"); cfgs = importlib.import_module("cal_22-mpc_amd.configs")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wl = sys.argv[2] if len(sys.argv) > 2 else "random_u32"
cfg = cfgs.probe_config(L)
scan = cfgs.plane_major_scan(L)
scan["Rows"][0], scan["Rows"][100] = scan["Rows"][100], scan["Rows"][0]      # a non-identity scan table
cfg["modules"]["3"]["submodules"]["ScanModule"] = scan
n = (1 << 30) // L
buf = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
mpc.synth_fill(buf.data_ptr(), n, L, wl)
torch.cuda.synchronize()
ev = mpc.VPC(cfg)
assert ev.kernel_path == mpc.MPC_PATH_VPC_GENERIC
st = torch.cuda.Stream()
ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(st)
for _ in range(3):
    ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
b.record(st)
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 3
print(f"generic VPC L={L} {wl}: {ms:.2f} ms / GiB = {1.0737 / ms * 1e3:.1f} GB/s, {n / ms / 1e3:.1f} M lines/s")
