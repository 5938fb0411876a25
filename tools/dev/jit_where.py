import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mpc = importlib.import_module("cal_22-mpc_amd"); C = importlib.import_module("cal_22-mpc_amd.configs")
L = 64
cfg = C.make_config(L, [{"name": "AllZero"}, C.consecutive_base(L, 0, True), C.one_base(L, 9, True)])
os.environ["MPC_JIT_CACHE"] = ""
ev = mpc.VPC(cfg)
print("form:", ev.kernel_form)
