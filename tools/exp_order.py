import importlib, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
mpc = importlib.import_module("cal_22-mpc_amd")
configs = importlib.import_module("cal_22-mpc_amd.configs"); traces = importlib.import_module("cal_22-mpc_amd.traces")
b = mpc.VPC(configs.probe_config(64))
s, c = b.compress_lines(traces.random_u32(1000))
print("my lib ok, device", b.info.device, np.unique(s), flush=True)
import torch
print("torch", torch.__version__, "avail", torch.cuda.is_available(), flush=True)
x = torch.zeros(4, device="cuda:0"); print("torch cuda tensor ok", x.sum().item())
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    n = 1 << 20
    buf = torch.empty(n * 64, dtype=torch.uint8, device="cuda:0")
    mpc.synth_fill(buf.data_ptr(), n, 64, "random_u32", stream=st.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st); b.compress_device(buf.data_ptr(), n, stream=st.cuda_stream); e1.record(st)
    torch.cuda.synchronize()
    print("kernel ms on torch stream", e0.elapsed_time(e1), "lines", int(b.stats_vector()[0]))
os.system("grep -E 'amdhip|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
