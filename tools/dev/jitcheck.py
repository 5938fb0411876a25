import importlib, sys
import os; sys_path = __import__("sys").path; sys_path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
mpc = importlib.import_module("cal_22-mpc_amd"); configs = importlib.import_module("cal_22-mpc_amd.configs")
az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}
for L in (32, 64, 128):
    def trunc(ts):
        return {"TableSize": ts, "Rows": [i // L for i in range(ts)], "Cols": [i % L for i in range(ts)]}
    prev1 = [max(i - 1, 0) for i in range(L)]
    prev4 = [max(i - 4, 0) for i in range(L)]
    diff = [(-2 + (i % 5)) for i in range(L)]
    w2 = [[1.0, 0.5][i % 2] for i in range(L)]
    for root in (1, 3, 4, 7, L // 2 + 1, L - 1):
        mods = [az, aws, configs.one_base(L, root, True), configs.diff_base(L, prev1, diff, root, False),
                configs.weight_base(L, prev4, w2, root, True), configs.one_base(L, 0, False)]
        cfg = configs.make_config(L, mods)
        d = mpc.describe_config(cfg)
        print(L, "root", root, d["sequence"], d["compiled"], d["general_layout"], flush=True)
        print("   code", mpc.jit_compile_check(cfg), flush=True)
    for ts in (8 * L - 24, 6 * L, 4 * L + 7, L, 16, 0):
        mods = [az, aws, configs.one_base(L, 0, True, trunc(ts)), configs.consecutive_base(L, 0, False, trunc(ts)),
                configs.diff_base(L, prev4, diff, 0, True, trunc(ts)), configs.weight_base(L, prev4, w2, 0, True, trunc(ts))]
        cfg = configs.make_config(L, mods)
        d = mpc.describe_config(cfg)
        print(L, "ts", ts, d["sequence"], d["compiled"], d["general_layout"], flush=True)
        print("   code", mpc.jit_compile_check(cfg), flush=True)
