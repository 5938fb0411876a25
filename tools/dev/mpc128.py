import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mpc = importlib.import_module("cal_22-mpc_amd"); C = importlib.import_module("cal_22-mpc_amd.configs")
L = 128; n = (8 << 30) // L
st = torch.cuda.Stream(); os.environ["MPC_JIT_CACHE"] = ""
for wl in ("pointers_u64", "random_u32", "mixed"):
    buf = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
    mpc.synth_fill(buf.data_ptr(), n, L, wl); torch.cuda.synchronize()
    for name, cfg in (("mpc_config(128)", C.mpc_config(L)), ("probe_config(128)", C.probe_config(L))):
        for jit in ("1", "0"):
            os.environ["MPC_JIT"] = jit
            ev = mpc.VPC(cfg)
            for _ in range(3): ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            for _ in range(5): ev.compress_device(buf.data_ptr(), n, stream=st.cuda_stream)
            b.record(st); torch.cuda.synchronize()
            print(f"{wl:13s} {name:18s} MPC_JIT={jit} {ev.kernel_form:32s} {n * L / (a.elapsed_time(b) / 5) / 1e6 / 8000:.3f} of peak", flush=True)
            ev.close()
