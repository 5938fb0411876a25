#!/usr/bin/env python3
"""Generate tests/golden/random_u32_32Mi_exceptions.json: the lines among the first
32 Mi lines of the random-uint32 workload (BASELINE.json configs[1], seed 12345)
that the CPU oracle does NOT size at 515 bits / cluster -1.  The GPU full-size
test checks its per-line outputs against this list (about 2 minutes on 8 cores)."""
import importlib
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
cfgs = importlib.import_module("cal_22-mpc_amd.configs")
tr = importlib.import_module("cal_22-mpc_amd.traces")
from oracle import oracle as O  # noqa: E402

N, T = 32 << 20, 8
per = N // T
res = [None] * T


def work(i):
    o = O.VpcOracle(cfgs.probe_config(64))
    bad = []
    for c in range(0, per, 1 << 18):
        m = min(1 << 18, per - c)
        s, sel = o.compress(tr.random_u32(m, 64, first_line=i * per + c), stats=False)
        for j in np.nonzero((s != 515) | (sel != -1))[0]:
            bad.append({"line": i * per + c + int(j), "size": int(s[j]), "cluster": int(sel[j])})
    res[i] = bad


ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
[t.start() for t in ths]
[t.join() for t in ths]
out = {"provenance": "CPU oracle (oracle/mpc_oracle.c, probe_config(64)) over traces.random_u32(32<<20, 64)",
       "n_lines": N, "default_size": 515, "default_cluster": -1,
       "exceptions": sorted(sum(res, []), key=lambda e: e["line"])}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "random_u32_32Mi_exceptions.json"), "w"), indent=1)
print(len(out["exceptions"]), "exceptions")
