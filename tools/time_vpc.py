#!/usr/bin/env python3
"""Development helper: kernel time of the VPC evaluator for a line size / workload (device-resident,
16 GiB), honouring MPC_HIP_LIB.   python tools/time_vpc.py L workload [algo]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpc = importlib.import_module("cal_22-mpc_amd"); cfgs = importlib.import_module("cal_22-mpc_amd.configs")
L = int(sys.argv[1]); wl = sys.argv[2]; algo = sys.argv[3] if len(sys.argv) > 3 else "VPC"
n = (16 << 30) // L
buf = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
mpc.synth_fill(buf.data_ptr(), n, L, wl)
torch.cuda.synchronize()
ev = mpc.VPC(cfgs.probe_config(L)) if algo == "VPC" else {"BDI": mpc.BDI, "FPC": mpc.FPC, "BPC": mpc.BPC}[algo](L)
st = torch.cuda.Stream()
for _ in range(2):
    ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(st)
for _ in range(5):
    ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
b.record(st)
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print(f"{os.path.basename(os.environ.get('MPC_HIP_LIB', 'default'))} {algo} L={L} {wl}: {ms:.3f} ms / 16 GiB = {17.18 / ms * 1e3:.0f} GB/s ({17.18 / ms * 1e3 / 8000:.3f} of peak)")
