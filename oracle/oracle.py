"""ctypes front-end of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  It loads ``oracle/libmpc_oracle.so`` (built by
``oracle/Makefile`` from ``mpc_oracle.c``) and, when present,
``oracle/_ref/libmpc_refstages.so`` (the reference's own stage classes).

The JSON -> struct translation below follows ``VPC::parseConfig`` (reference
``src/compressor/VPC.cpp:72-330``) using Python's ``json``.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import subprocess
from typing import Dict, Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LINE = 256
MAX_MODULES = 32
MAX_TABLE = 8 * MAX_LINE

KIND_ALLZERO, KIND_ALLWORDSAME, KIND_PREDCOMP = 0, 1, 2
PRED = {"WeightBasePredictor": 0, "DiffBasePredictor": 1, "OneBasePredictor": 2,
        "ConsecutiveBasePredictor": 3}


class OModule(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("pred_kind", C.c_int32), ("root", C.c_int32),
        ("consecutive_xor", C.c_int32), ("table_size", C.c_int32),
        ("base", C.c_int32 * MAX_LINE), ("weight", C.c_float * MAX_LINE),
        ("diff", C.c_int32 * MAX_LINE),
        ("rows", C.c_int32 * MAX_TABLE), ("cols", C.c_int32 * MAX_TABLE),
    ]


class OConfig(C.Structure):
    _fields_ = [
        ("num_modules", C.c_int32), ("line_size", C.c_int32),
        ("enc_bits", C.c_int32 * (MAX_MODULES + 1)),
        ("modules", OModule * MAX_MODULES),
    ]


K = MAX_MODULES + 1


class OVpcStats(C.Structure):
    _fields_ = [
        ("lines", C.c_uint64), ("original_bits", C.c_uint64), ("compressed_bits", C.c_uint64),
        ("comp_ratio", C.c_double),
        ("count", C.c_uint64 * K), ("c_original_bits", C.c_uint64 * K),
        ("c_compressed_bits", C.c_uint64 * K), ("c_comp_ratio", C.c_double * K),
        ("sum_mae", C.c_double * K), ("sum_mse", C.c_double * K),
        ("mae", C.c_double * K), ("mse", C.c_double * K),
        ("residue_lines", C.c_uint64 * K),
        ("sum_r", C.c_uint64 * K), ("sum_r2", C.c_uint64 * K),
        ("hist", C.POINTER(C.c_uint64)), ("hist_bins", C.c_uint32),
    ]


class OBdiStats(C.Structure):
    _fields_ = [
        ("lines", C.c_uint64), ("original_bits", C.c_uint64), ("compressed_bits", C.c_uint64),
        ("comp_ratio", C.c_double), ("counts", C.c_uint64 * 9),
    ]


class OFpcStats(C.Structure):
    _fields_ = [
        ("lines", C.c_uint64), ("original_bits", C.c_uint64), ("compressed_bits", C.c_uint64),
        ("comp_ratio", C.c_double), ("total_words", C.c_uint64), ("counts", C.c_uint64 * 8),
    ]


class OBpcStats(C.Structure):
    _fields_ = [
        ("lines", C.c_uint64), ("original_bits", C.c_uint64), ("compressed_bits", C.c_uint64),
        ("comp_ratio", C.c_double), ("total_words", C.c_uint64), ("counts", C.c_uint64 * 7),
    ]


def build(ref: bool = True) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is mounted)."""
    subprocess.run(["make", "-s", "-C", HERE, "all"], check=True)
    if ref and os.path.isdir("/root/reference/src/compressor/VPCmodules"):
        subprocess.run(["make", "-s", "-C", HERE, "_ref"], check=True)


_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libmpc_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.mpc_o_vpc_validate.argtypes = [C.POINTER(OConfig)]
        L.mpc_o_vpc_validate.restype = C.c_int
        L.mpc_o_vpc_line.argtypes = [C.POINTER(OConfig), C.c_void_p, C.POINTER(C.c_int),
                                     C.POINTER(OVpcStats)]
        L.mpc_o_vpc_line.restype = C.c_uint
        L.mpc_o_vpc_batch.argtypes = [C.POINTER(OConfig), C.c_void_p, C.c_uint64, C.c_void_p,
                                      C.c_void_p, C.POINTER(OVpcStats)]
        L.mpc_o_vpc_batch.restype = None
        L.mpc_o_predict.argtypes = [C.POINTER(OModule), C.c_int, C.c_void_p, C.c_void_p]
        L.mpc_o_residue.argtypes = [C.POINTER(OModule), C.c_int, C.c_void_p, C.c_void_p]
        L.mpc_o_scanned.argtypes = [C.POINTER(OModule), C.c_int, C.c_void_p, C.c_void_p]
        L.mpc_o_fpc_size.argtypes = [C.c_void_p, C.c_int]
        L.mpc_o_fpc_size.restype = C.c_int
        L.mpc_o_bdi_batch.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p,
                                      C.POINTER(OBdiStats)]
        L.mpc_o_bdi_batch.restype = None
        L.mpc_o_fpc_batch.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.POINTER(OFpcStats)]
        L.mpc_o_fpc_batch.restype = None
        L.mpc_o_bpc_batch.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.POINTER(OBpcStats)]
        L.mpc_o_bpc_batch.restype = None
        L.mpc_o_bdi_reduce_sign.argtypes = [C.c_uint64]
        L.mpc_o_bdi_reduce_sign.restype = C.c_uint64
        L.mpc_o_bdi_check.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_uint]
        L.mpc_o_bdi_check.restype = C.c_uint
        _lib = L
    return _lib


def ref_lib() -> Optional[C.CDLL]:
    """The reference's own stage classes (oracle/_ref), or None if not built."""
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libmpc_refstages.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        pred_args = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_predict.argtypes = pred_args + [C.c_void_p, C.c_void_p]
        R.ref_residue.argtypes = pred_args + [C.c_void_p, C.c_void_p]
        R.ref_mae_mse.argtypes = pred_args + [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        R.ref_scanned.argtypes = pred_args + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_fpc_size.argtypes = [C.c_void_p, C.c_int]
        R.ref_fpc_size.restype = C.c_int
        _ref = R
    return _ref


# --------------------------------------------------------------------------
# JSON -> OConfig  (VPC.cpp:72-330)
# --------------------------------------------------------------------------

def config_from_json(cfg: Dict) -> OConfig:
    oc = OConfig()
    ov = cfg["overview"]
    M = int(ov["num_modules"])
    L = int(ov["lineSize"])
    oc.num_modules, oc.line_size = M, L
    if ov.get("encoding_bits") is None:
        # (int)ceil(log2f((float)m_NumClusters)), VPC.cpp:104
        eb = int(math.ceil(math.log2(float(M + 1))))
        for k in range(M + 1):
            oc.enc_bits[k] = eb
    else:
        for k in range(M + 1):
            oc.enc_bits[k] = int(ov["encoding_bits"][k])
    for i in range(M):
        spec = cfg["modules"][str(i)]
        m = oc.modules[i]
        name = spec["name"]
        if name == "AllZero":
            m.kind = KIND_ALLZERO
        elif name in ("AllWordSame", "ByteplaneAllSame"):
            m.kind = KIND_ALLWORDSAME
        elif name == "PredComp":
            m.kind = KIND_PREDCOMP
            sub = spec["submodules"]
            p = sub["ResidueModule"]["PredictorModule"]
            m.pred_kind = PRED[p["name"]]
            pl = int(p["LineSize"])
            if pl != L:
                raise ValueError("predictor LineSize != overview.lineSize is not supported")
            m.root = int(p["RootIndex"])
            if m.pred_kind in (0, 1):
                for j in range(L):
                    m.base[j] = int(p["BaseIndexTable"][j])
                    if m.pred_kind == 0:
                        m.weight[j] = float(p["WeightTable"][j])
                    else:
                        m.diff[j] = int(p["DiffTable"][j])
            m.consecutive_xor = 1 if sub["XORModule"]["consecutiveXOR"] else 0
            sc = sub["ScanModule"]
            m.table_size = int(sc["TableSize"])
            if m.table_size > MAX_TABLE:
                raise ValueError("scan table too large")
            for j in range(m.table_size):
                m.rows[j] = int(sc["Rows"][j])
                m.cols[j] = int(sc["Cols"][j])
        else:
            raise ValueError(f"invalid module {name}")
    return oc


def hist_bins(oc: OConfig) -> int:
    eb = max(oc.enc_bits[k] for k in range(oc.num_modules + 1))
    return max(288, 8 * oc.line_size + eb + 1)


class VpcOracle:
    """Stateful evaluator mirroring ``comp::VPC`` + ``VPCResult``."""

    def __init__(self, cfg: Dict):
        self.cfg_json = cfg
        self.oc = config_from_json(cfg)
        rc = lib().mpc_o_vpc_validate(C.byref(self.oc))
        if rc != 0:
            raise ValueError(f"configuration rejected by the oracle ({rc})")
        self.L = self.oc.line_size
        self.M = self.oc.num_modules
        self.bins = hist_bins(self.oc)
        self.reset()

    def reset(self) -> None:
        self.st = OVpcStats()
        self._hist = np.zeros((MAX_MODULES + 1) * self.bins, dtype=np.uint64)
        self.st.hist = self._hist.ctypes.data_as(C.POINTER(C.c_uint64))
        self.st.hist_bins = self.bins

    def compress(self, lines: np.ndarray, stats: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        lines = np.ascontiguousarray(lines, dtype=np.uint8)
        assert lines.ndim == 2 and lines.shape[1] == self.L
        n = lines.shape[0]
        sizes = np.zeros(n, dtype=np.uint16)
        sel = np.zeros(n, dtype=np.int8)
        lib().mpc_o_vpc_batch(C.byref(self.oc), lines.ctypes.data, n, sizes.ctypes.data,
                              sel.ctypes.data, C.byref(self.st) if stats else None)
        return sizes, sel

    def stats_vector(self) -> np.ndarray:
        """The integer statistics vector in the layout of include/mpc_hip.h
        (``mpc_stats_get``)."""
        M, B = self.M, self.bins
        v = [self.st.lines, self.st.original_bits, self.st.compressed_bits]
        for k in range(M + 1):
            v += [self.st.count[k], self.st.c_original_bits[k], self.st.c_compressed_bits[k],
                  self.st.residue_lines[k], self.st.sum_r[k], self.st.sum_r2[k]]
        out = np.array(v, dtype=np.uint64)
        h = self._hist.reshape(MAX_MODULES + 1, B)[: M + 1].reshape(-1)
        return np.concatenate([out, h])

    def hist(self) -> np.ndarray:
        return self._hist.reshape(MAX_MODULES + 1, self.bins)[: self.M + 1].copy()


class BdiOracle:
    def __init__(self, line_size: int):
        if line_size % 8 or line_size < 8 or line_size > MAX_LINE:
            raise ValueError("BDI needs a line size that is a multiple of 8")
        self.L = line_size
        self.reset()

    def reset(self) -> None:
        self.st = OBdiStats()

    def compress(self, lines: np.ndarray, stats: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        lines = np.ascontiguousarray(lines, dtype=np.uint8)
        assert lines.ndim == 2 and lines.shape[1] == self.L
        n = lines.shape[0]
        sizes = np.zeros(n, dtype=np.uint16)
        sel = np.zeros(n, dtype=np.int8)
        lib().mpc_o_bdi_batch(lines.ctypes.data, self.L, n, sizes.ctypes.data, sel.ctypes.data,
                              C.byref(self.st) if stats else None)
        return sizes, sel

    def stats_vector(self) -> np.ndarray:
        return np.array([self.st.lines, self.st.original_bits, self.st.compressed_bits]
                        + [self.st.counts[i] for i in range(9)], dtype=np.uint64)


class FpcOracle:
    """FPC::CompressLine restatement (reference FPC.cpp:7-88) -- PARITY UNPINNED, see mpc_oracle.h."""

    def __init__(self, line_size: int):
        if line_size % 4 or line_size < 4 or line_size > MAX_LINE:
            raise ValueError("FPC needs a line size that is a multiple of 4")
        self.L = line_size
        self.reset()

    def reset(self) -> None:
        self.st = OFpcStats()

    def compress(self, lines: np.ndarray, stats: bool = True) -> np.ndarray:
        lines = np.ascontiguousarray(lines, dtype=np.uint8)
        assert lines.ndim == 2 and lines.shape[1] == self.L
        n = lines.shape[0]
        sizes = np.zeros(n, dtype=np.uint16)
        lib().mpc_o_fpc_batch(lines.ctypes.data, self.L, n, sizes.ctypes.data, C.byref(self.st) if stats else None)
        return sizes

    def stats_vector(self) -> np.ndarray:
        return np.array([self.st.lines, self.st.original_bits, self.st.compressed_bits]
                        + [self.st.counts[i] for i in range(8)], dtype=np.uint64)


class BpcOracle:
    """BPC::CompressLine restatement (reference BPC.cpp:20-185) -- PARITY UNPINNED, see mpc_oracle.h."""

    def __init__(self, line_size: int):
        if line_size % 4 or line_size < 8 or line_size > 128:
            raise ValueError("BPC needs 8..128-byte lines, a multiple of 4")
        self.L = line_size
        self.reset()

    def reset(self) -> None:
        self.st = OBpcStats()

    def compress(self, lines: np.ndarray, stats: bool = True) -> np.ndarray:
        lines = np.ascontiguousarray(lines, dtype=np.uint8)
        assert lines.ndim == 2 and lines.shape[1] == self.L
        n = lines.shape[0]
        sizes = np.zeros(n, dtype=np.uint16)
        lib().mpc_o_bpc_batch(lines.ctypes.data, self.L, n, sizes.ctypes.data, C.byref(self.st) if stats else None)
        return sizes

    def stats_vector(self) -> np.ndarray:
        return np.array([self.st.lines, self.st.original_bits, self.st.compressed_bits, self.st.total_words]
                        + [self.st.counts[i] for i in range(7)], dtype=np.uint64)

