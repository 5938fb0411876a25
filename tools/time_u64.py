import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
mpc = importlib.import_module("cal_22-mpc_amd"); cfgs = importlib.import_module("cal_22-mpc_amd.configs")
L = 128; n = (16 << 30) // L
buf = torch.empty(n * L, dtype=torch.uint8, device="cuda:0"); mpc.synth_fill(buf.data_ptr(), n, L, "pointers_u64"); torch.cuda.synchronize()
for name, cfg in (("probe (i-4)", cfgs.probe_config(L)), ("probe_u64 (i-8)", cfgs.probe_config_u64(L))):
    ev = mpc.VPC(cfg); st = torch.cuda.Stream()
    for _ in range(2): ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
    torch.cuda.synchronize(); ev.reset()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(5): ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
    b.record(st); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"{name}: path {ev.kernel_path} {ms:.3f} ms / 16 GiB = {17.18/ms*1e3:.0f} GB/s ({17.18/ms*1e3/8000:.3f}), ratio {ev.result()['comp_ratio']:.4f}")
