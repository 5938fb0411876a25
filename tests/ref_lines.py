"""Shared by the CPU and GPU tests: what the selector / decision rule of the reference
(VPC.cpp:366-415) gives when it is applied to numbers produced by the reference's OWN stage
classes (tests/golden/ref_line_vectors.json, written by tests/golden/make_ref_line_vectors.py)."""
import hashlib
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _generator():
    spec = importlib.util.spec_from_file_location("make_ref_line_vectors",
                                                  os.path.join(HERE, "golden", "make_ref_line_vectors.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)     # only defines the case table; main() is what needs /root/reference
    return mod


def load_cases():
    """[(name, cfg, lines, fixture)] with the regenerated inputs checked against the stored checksums."""
    with open(os.path.join(HERE, "golden", "ref_line_vectors.json")) as f:
        fx = json.load(f)["cases"]
    gen = _generator()
    out = []
    for name, (mk_cfg, mk_lines) in gen.CASES.items():
        c = fx[name]
        cfg, lines = mk_cfg(), np.ascontiguousarray(mk_lines())
        assert hashlib.sha256(lines.tobytes()).hexdigest() == c["lines_sha256"], name
        assert hashlib.sha256(json.dumps(cfg, sort_keys=True).encode()).hexdigest() == c["config_sha256"], name
        out.append((name, cfg, lines, c))
    assert len(out) == len(fx)
    return out


def expected_sizes(oracle_mod, cfg, lines, c):
    """(size, selected) per line from the reference-produced z / enc numbers."""
    oc = oracle_mod.config_from_json(cfg)
    L, M = oc.line_size, oc.num_modules
    eb = [oc.enc_bits[k] for k in range(M + 1)]           # index cluster + 1
    has_aws = M > 1 and oc.modules[1].kind == oracle_mod.KIND_ALLWORDSAME
    z, enc, idx = np.array(c["z"]), np.array(c["enc"]), c["module_index"]
    n = len(lines)
    size = np.zeros(n, dtype=np.uint16)
    sel = np.zeros(n, dtype=np.int8)
    words = lines.reshape(n, L // 4, 4)
    is_zero = ~lines.any(axis=1)                                          # VPC.cpp:332-347
    is_same = (words == words[:, :1, :]).all(axis=(1, 2))                 # VPC.cpp:349-364
    for i in range(n):
        if is_zero[i]:
            size[i], sel[i] = eb[1], 0
            continue
        if has_aws and is_same[i]:
            size[i], sel[i] = 32 + eb[2], 1
            continue
        best_z, q_best = 0, -1
        for q in range(len(idx)):                                         # VPC.cpp:377-395, ties -> later module
            if best_z <= z[i, q]:
                best_z, q_best = z[i, q], q
        e = int(enc[i, q_best])
        if e < 8 * L:                                                     # VPC.cpp:397-407
            size[i], sel[i] = e + eb[idx[q_best] + 1], idx[q_best]
        else:
            size[i], sel[i] = 8 * L + eb[0], -1
    return size, sel
