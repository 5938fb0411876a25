"""VPC JSON configuration authoring.

The reference ships no sample configuration; its schema is defined only by
``VPC::parseConfig`` (reference ``src/compressor/VPC.cpp:72-330``).  This module
writes configurations in exactly that schema:

``overview``: ``num_modules`` (M), ``lineSize`` (L), optional ``encoding_bits``
(M+1 ints, element k -> cluster k-1, ``VPC.cpp:111-115``).
``modules``: object keyed ``"0"``..``"M-1"``; ``name`` in ``AllZero`` |
``AllWordSame``/``ByteplaneAllSame`` | ``PredComp``; ``PredComp.submodules`` =
``ResidueModule.PredictorModule{name, LineSize, RootIndex, [BaseIndexTable,
WeightTable | DiffTable]}``, ``XORModule.consecutiveXOR``,
``ScanModule{TableSize, Rows, Cols}``, ``FPCModule{num_modules, ...}``
(parsed by the reference, then unused: ``VPC.cpp:303-306``).
"""
from __future__ import annotations

import json
from typing import Dict, List, Optional


def plane_major_scan(line_size: int) -> Dict:
    """Identity plane-major scan: scanned bit i = XORed plane i // L, column i % L
    (``ScanModule.cpp:13-19``)."""
    n = 8 * line_size
    return {
        "TableSize": n,
        "Rows": [i // line_size for i in range(n)],
        "Cols": [i % line_size for i in range(n)],
    }


def _pred_comp(pred: Dict, consecutive_xor: bool, scan: Dict) -> Dict:
    return {
        "name": "PredComp",
        "submodules": {
            "ResidueModule": {"PredictorModule": pred},
            "XORModule": {"consecutiveXOR": bool(consecutive_xor)},
            "ScanModule": scan,
            "FPCModule": {"num_modules": 0},
        },
    }


def one_base(line_size: int, root: int = 0, consecutive_xor: bool = True, scan=None) -> Dict:
    pred = {"name": "OneBasePredictor", "LineSize": line_size, "RootIndex": root}
    return _pred_comp(pred, consecutive_xor, scan or plane_major_scan(line_size))


def consecutive_base(line_size: int, root: int = 0, consecutive_xor: bool = True, scan=None) -> Dict:
    pred = {"name": "ConsecutiveBasePredictor", "LineSize": line_size, "RootIndex": root}
    return _pred_comp(pred, consecutive_xor, scan or plane_major_scan(line_size))


def diff_base(line_size: int, base: List[int], diff: List[int], root: int = 0,
              consecutive_xor: bool = False, scan=None) -> Dict:
    assert len(base) == line_size and len(diff) == line_size
    pred = {"name": "DiffBasePredictor", "LineSize": line_size, "RootIndex": root,
            "BaseIndexTable": list(map(int, base)), "DiffTable": list(map(int, diff))}
    return _pred_comp(pred, consecutive_xor, scan or plane_major_scan(line_size))


def weight_base(line_size: int, base: List[int], weight: List[float], root: int = 0,
                consecutive_xor: bool = True, scan=None) -> Dict:
    assert len(base) == line_size and len(weight) == line_size
    pred = {"name": "WeightBasePredictor", "LineSize": line_size, "RootIndex": root,
            "BaseIndexTable": list(map(int, base)), "WeightTable": list(map(float, weight))}
    return _pred_comp(pred, consecutive_xor, scan or plane_major_scan(line_size))


def make_config(line_size: int, modules: List[Dict], encoding_bits: Optional[List[int]] = None) -> Dict:
    cfg = {
        "overview": {"num_modules": len(modules), "lineSize": line_size},
        "modules": {str(i): m for i, m in enumerate(modules)},
    }
    if encoding_bits is not None:
        assert len(encoding_bits) == len(modules) + 1
        cfg["overview"]["encoding_bits"] = list(map(int, encoding_bits))
    return cfg


def probe_config(line_size: int = 64, encoding_bits: Optional[List[int]] = None) -> Dict:
    """The 6-module configuration SURVEY.md 8c quotes known answers for:
    AllZero, AllWordSame, OneBase(consecutive XOR), ConsecutiveBase(consecutive
    XOR), DiffBase(base[i]=max(i-4,0), diff 1 on word-LSBs, plane-0 XOR),
    WeightBase(base[i]=max(i-4,0), weight 1 / 0.5 on even / odd bytes,
    consecutive XOR); identity plane-major scan; default id bits."""
    L = line_size
    base = [max(i - 4, 0) for i in range(L)]
    diff = [1 if i % 4 == 0 else 0 for i in range(L)]
    weight = [1.0 if i % 2 == 0 else 0.5 for i in range(L)]
    mods = [
        {"name": "AllZero"},
        {"name": "AllWordSame"},
        one_base(L, 0, True),
        consecutive_base(L, 0, True),
        diff_base(L, base, diff, 0, False),
        weight_base(L, base, weight, 0, True),
    ]
    return make_config(L, mods, encoding_bits)


def probe_config_u64(line_size: int = 64, encoding_bits: Optional[List[int]] = None) -> Dict:
    """The probe configuration for 8-byte elements (pointers, int64, fp64): the DiffBase /
    WeightBase predictors look two words back (BaseIndexTable[i] = max(i-8, 0)); diff 1 on the
    least significant byte of every element, weights 1 / 0.5 on even / odd bytes."""
    L = line_size
    base = [max(i - 8, 0) for i in range(L)]
    diff = [1 if i % 8 == 0 else 0 for i in range(L)]
    weight = [1.0 if i % 2 == 0 else 0.5 for i in range(L)]
    mods = [
        {"name": "AllZero"},
        {"name": "AllWordSame"},
        one_base(L, 0, True),
        consecutive_base(L, 0, True),
        diff_base(L, base, diff, 0, False),
        weight_base(L, base, weight, 0, True),
    ]
    return make_config(L, mods, encoding_bits)


def element_config(line_size: int, element_bytes: int, encoding_bits: Optional[List[int]] = None) -> Dict:
    """Configuration authoring helper (the reference ships no configuration files): the probe
    module set -- AllZero, AllWordSame, OneBase, ConsecutiveBase, DiffBase, WeightBase -- with the
    Diff / Weight predictors looking one element back, for 1-, 2-, 4- or 8-byte elements
    (``BaseIndexTable[i] = max(i - element_bytes, 0)``, diff 1 on each element's least
    significant byte, weight 1 / 0.5 on even / odd bytes).  Every such configuration runs on the
    fast kernel path."""
    if element_bytes not in (1, 2, 4, 8):
        raise ValueError("element_bytes must be 1, 2, 4 or 8")
    L = line_size
    base = [max(i - element_bytes, 0) for i in range(L)]
    diff = [1 if i % element_bytes == 0 else 0 for i in range(L)]
    weight = [1.0 if i % 2 == 0 else 0.5 for i in range(L)]
    mods = [{"name": "AllZero"}, {"name": "AllWordSame"}, one_base(L, 0, True), consecutive_base(L, 0, True),
            diff_base(L, base, diff, 0, False), weight_base(L, base, weight, 0, True)]
    return make_config(L, mods, encoding_bits)


# ---- per-datatype prediction models (the five models of the paper's overview figure, MPC.PNG) ----
# The reference ships no configuration files, so the TABLES below are this repository's: one prediction
# module per data type in the figure -- Bool/INT8, INT16, INT32/64, FP32, FP64 -- each predicting a byte
# from the same byte of the previous element (DiffBase), the IEEE formats with the exponent bytes at full
# weight and mantissa bytes at half weight (WeightBase) or through the byte-plane shuffle
# (ConsecutiveBase, 32-bit words).  Every one of them runs on the fast kernel path.
DTYPE_BYTES = {"bool": 1, "int8": 1, "int16": 2, "int32": 4, "int64": 8, "fp16": 2, "fp32": 4, "fp64": 8}


def datatype_module(line_size: int, dtype: str) -> Dict:
    """The prediction module for one data type (see above)."""
    L = line_size
    if dtype not in DTYPE_BYTES:
        raise ValueError(f"dtype must be one of {sorted(DTYPE_BYTES)}")
    eb = DTYPE_BYTES[dtype]
    base = [max(i - eb, 0) for i in range(L)]
    if dtype in ("bool", "int8", "int16", "int32", "int64"):
        # previous element, same byte; consecutive XOR turns small signed differences into short codes
        return diff_base(L, base, [0] * L, 0, True)
    if dtype == "fp32":
        return consecutive_base(L, 0, True)
    # fp16 / fp64: little-endian elements, sign + exponent in the top byte(s): full weight there, half below
    top = 1 if dtype == "fp16" else 2
    weight = [1.0 if (i % eb) >= eb - top else 0.5 for i in range(L)]
    return weight_base(L, base, weight, 0, True)


def datatype_config(line_size: int, dtype: str, encoding_bits: Optional[List[int]] = None) -> Dict:
    """AllZero, AllWordSame and the prediction module of one data type."""
    return make_config(line_size, [{"name": "AllZero"}, {"name": "AllWordSame"}, datatype_module(line_size, dtype)],
                       encoding_bits)


def mpc_config(line_size: int = 32, encoding_bits: Optional[List[int]] = None) -> Dict:
    """The seven-module shape of the paper's overview figure (MPC.PNG; 32-byte blocks there): all-zero and
    all-words-same tests and the five prediction models Bool/INT8, INT16, INT32/64, FP32, FP64."""
    L = line_size
    mods = [{"name": "AllZero"}, {"name": "AllWordSame"}, datatype_module(L, "int8"), datatype_module(L, "int16"),
            datatype_module(L, "int32"), datatype_module(L, "fp32"), datatype_module(L, "fp64")]
    return make_config(L, mods, encoding_bits)


def write_config(cfg: Dict, path: str) -> str:
    with open(path, "w") as f:
        json.dump(cfg, f)
    return path


if __name__ == "__main__":
    # python cal_22-mpc_amd/configs.py --line 64 --model mpc -o cfg.json
    import argparse
    ap = argparse.ArgumentParser(description="write a VPC configuration for the `compressor -c` option")
    ap.add_argument("--line", type=int, default=64, help="line size in bytes (32, 64 or 128 for the fast kernel)")
    ap.add_argument("--model", default="probe", choices=["probe", "mpc"] + sorted(DTYPE_BYTES),
                    help="probe: the 6-module probe set for --element-byte elements; mpc: the paper figure's five "
                         "data-type models; a data type: that type's model alone")
    ap.add_argument("--element", type=int, default=4, choices=[1, 2, 4, 8], help="element size in bytes (probe model)")
    ap.add_argument("-o", "--output", default="-")
    a = ap.parse_args()
    cfg = element_config(a.line, a.element) if a.model == "probe" else mpc_config(a.line) if a.model == "mpc" \
        else datatype_config(a.line, a.model)
    text = json.dumps(cfg)
    if a.output == "-":
        print(text)
    else:
        with open(a.output, "w") as f:
            f.write(text)
