"""MI355X-native per-cache-line multi-prediction compression evaluator.

Python is only a thin ctypes binding over the C ABI of ``include/mpc_hip.h``
(``libmpc_hip.so``: hand-written gfx950 kernels + the pinned double-buffered
stager).  There is no Python or CPU evaluation path here: if the native library
is missing or no HIP device is present, construction raises.

The classes mirror the reference's operator interface for this path
(``comp::VPC`` / ``comp::BDI`` behind ``comp::Compressor``; reference
``src/compressor/Compressor.h:18-33``, ``VPC.h:241-283``, ``BDI.h:90-107``):
``compress_lines`` is the batch form of ``CompressLine`` and ``result()`` the
``GetResult()`` statistics.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPC_HIP_LIB") or os.path.join(HERE, "libmpc_hip.so")   # override: development builds only

MPC_PATH_VPC_FAST, MPC_PATH_VPC_GENERIC, MPC_PATH_BDI, MPC_PATH_FPC, MPC_PATH_BPC = 1, 2, 3, 4, 5
SYNTH_KINDS = {"zeros": 0, "random_u32": 1, "sine_f32": 2, "mixed": 3, "pointers_u64": 4}


class MpcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mpc error {code}: {msg}")
        self.code = code


class Info(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("algorithm", C.c_int32), ("line_size", C.c_int32),
                ("num_modules", C.c_int32), ("num_clusters", C.c_int32), ("hist_bins", C.c_int32),
                ("kernel_path", C.c_int32), ("device", C.c_int32), ("stats_len", C.c_uint64)]


_lib = None


def _share_host_hip_runtime() -> None:
    """One HIP runtime per process.  The PyTorch wheel bundles its own
    libamdhip64.so / libhsa-runtime64.so next to libtorch_hip.so; if libmpc_hip.so
    brought up the system copy (/opt/rocm) as well, the process would hold two HSA
    runtimes, PyTorch could no longer see the GPU, and streams / device pointers
    could not be shared.  Promoting PyTorch's copy to the global symbol scope
    BEFORE libmpc_hip.so is bound (RTLD_NOW) makes every hip* reference of
    libmpc_hip.so resolve to that one runtime.  Without PyTorch (the C++ CLI)
    libmpc_hip.so simply uses the system runtime it is linked against."""
    try:
        import torch  # noqa: F401
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib() -> C.CDLL:
    """Load libmpc_hip.so (in-tree).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python {HERE}/build.py` "
                              "(there is no fallback implementation)")
        _share_host_hip_runtime()
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL | os.RTLD_NOW)
        H = C.c_void_p
        sigs = {
            "mpc_create_vpc": ([C.c_char_p, C.c_int, C.POINTER(H)], C.c_int),
            "mpc_create_vpc_from_string": ([C.c_char_p, C.c_int, C.POINTER(H)], C.c_int),
            "mpc_create_bdi": ([C.c_uint, C.c_int, C.POINTER(H)], C.c_int),
            "mpc_create_fpc": ([C.c_uint, C.c_int, C.POINTER(H)], C.c_int),
            "mpc_create_bpc": ([C.c_uint, C.c_int, C.POINTER(H)], C.c_int),
            "mpc_destroy": ([H], None),
            "mpc_get_info": ([H, C.POINTER(Info)], C.c_int),
            "mpc_last_error": ([H], C.c_char_p),
            "mpc_path_reason": ([H], C.c_char_p),
            "mpc_kernel_form": ([H], C.c_char_p),
            "mpc_jit_compile_check": ([C.c_char_p, C.c_char_p, C.c_size_t], C.c_longlong),
            "mpc_compress_batch": ([H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p], C.c_int),
            "mpc_compress_batch_device": ([H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
            "mpc_sync": ([H], C.c_int),
            "mpc_stats_len": ([H, C.POINTER(C.c_uint64)], C.c_int),
            "mpc_stats_get": ([H, C.c_void_p, C.c_size_t], C.c_int),
            "mpc_stats_merge": ([H, C.c_void_p, C.c_size_t], C.c_int),
            "mpc_stats_set": ([H, C.c_void_p, C.c_size_t], C.c_int),
            "mpc_stats_reset": ([H], C.c_int),
            "mpc_stats_raw_len": ([H, C.POINTER(C.c_uint64)], C.c_int),
            "mpc_stats_copy_raw_device": ([H, C.c_void_p, C.c_void_p], C.c_int),
            "mpc_stats_from_raw": ([H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t], C.c_int),
            "mpc_config_describe": ([C.c_char_p, C.c_char_p, C.c_size_t], C.c_int),
            "mpc_compress_npy": ([H, C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)], C.c_int),
            "mpc_npy_shape": ([C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)], C.c_int),
            "mpc_compress_gpgpusim_log": ([H, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)], C.c_int),
            "mpc_gpgpusim_log_line_size": ([C.c_char_p, C.POINTER(C.c_uint32)], C.c_int),
            "mpc_synth_fill": ([C.c_void_p, C.c_uint64, C.c_uint, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p], C.c_int),
            "mpc_read_bandwidth_probe": ([C.c_void_p, C.c_uint64, C.c_void_p], C.c_int),
        }
        for name, (args, res) in sigs.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
        _lib = L
    return _lib


EXPORTED_SYMBOLS = [
    "mpc_create_vpc", "mpc_create_vpc_from_string", "mpc_create_bdi", "mpc_create_fpc", "mpc_create_bpc", "mpc_destroy", "mpc_get_info",
    "mpc_last_error", "mpc_path_reason", "mpc_kernel_form", "mpc_jit_compile_check", "mpc_compress_batch", "mpc_compress_batch_device", "mpc_sync", "mpc_stats_len",
    "mpc_stats_get", "mpc_stats_merge", "mpc_stats_set", "mpc_stats_reset", "mpc_stats_raw_len",
    "mpc_stats_copy_raw_device", "mpc_stats_from_raw", "mpc_config_describe",
    "mpc_compress_npy", "mpc_npy_shape", "mpc_compress_gpgpusim_log", "mpc_gpgpusim_log_line_size",
    "mpc_synth_fill", "mpc_read_bandwidth_probe",
]


def gpgpusim_log_line_size(path: str) -> int:
    """Line size of a GPGPU-Sim ``.log`` trace: req_size of its first request (no device needed)."""
    sz = C.c_uint32()
    rc = lib().mpc_gpgpusim_log_line_size(path.encode(), C.byref(sz))
    if rc != 0:
        raise MpcError(rc, (lib().mpc_last_error(None) or b"").decode())
    return int(sz.value)


def describe_config(cfg) -> Dict:
    """Parse/validate a VPC configuration with the native parser (no device needed)."""
    text = cfg if isinstance(cfg, str) else json.dumps(cfg)
    buf = C.create_string_buffer(1 << 20)
    rc = lib().mpc_config_describe(text.encode(), buf, len(buf))
    out = json.loads(buf.value.decode())
    out["rc"] = rc
    return out


def jit_compile_check(cfg) -> int:
    """Build check of the run-time compilation (no device needed): when the configuration's module sequence would be compiled
    with hiprtc at handle creation, compile it now for gfx950 and return the code object's size; 0 when nothing would be
    compiled.  Raises MpcError with the compiler's log on failure."""
    text = cfg if isinstance(cfg, str) else json.dumps(cfg)
    log = C.create_string_buffer(1 << 16)
    n = lib().mpc_jit_compile_check(text.encode(), log, len(log))
    if n < 0:
        raise MpcError(int(n), log.value.decode(errors="replace"))
    return int(n)


class _Evaluator:
    """Common part of VPC / BDI: owns one ``mpc_handle``."""

    def __init__(self):
        self._h = C.c_void_p()
        self.info = Info()

    def _check(self, rc: int) -> None:
        if rc != 0:
            msg = lib().mpc_last_error(self._h if self._h else None)
            raise MpcError(rc, msg.decode() if msg else "")

    def _finish(self) -> None:
        self._check(lib().mpc_get_info(self._h, C.byref(self.info)))
        self.line_size = self.info.line_size
        self.stats_len = int(self.info.stats_len)

    def close(self) -> None:
        if self._h:
            lib().mpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- hot path -----------------------------------------------------------
    def compress_lines(self, lines: np.ndarray, want_sizes: bool = True,
                       want_selected: bool = True) -> Tuple[Optional[np.ndarray], Optional[np.ndarray]]:
        """Host buffer [n, L] uint8 -> (size_bits uint16[n], selected int8[n]); staged
        through the pinned double buffers.  Statistics accumulate in the handle."""
        lines = np.ascontiguousarray(lines, dtype=np.uint8)
        if lines.ndim != 2 or lines.shape[1] != self.line_size:
            raise ValueError(f"expected [n, {self.line_size}] uint8")
        n = lines.shape[0]
        sizes = np.empty(n, dtype=np.uint16) if want_sizes else None
        sel = np.empty(n, dtype=np.int8) if want_selected else None
        self._check(lib().mpc_compress_batch(self._h, lines.ctypes.data, n,
                                             sizes.ctypes.data if want_sizes else None,
                                             sel.ctypes.data if want_selected else None))
        return sizes, sel

    def compress_device(self, d_lines: int, n_lines: int, d_sizes: int = 0, d_selected: int = 0,
                        stream: int = 0) -> None:
        """Device-resident lines (raw pointers, e.g. ``tensor.data_ptr()``); asynchronous."""
        self._check(lib().mpc_compress_batch_device(self._h, d_lines, n_lines, d_sizes or None,
                                                    d_selected or None, stream or None))

    def compress_npy(self, path: str, first_row: int = 0, n_rows: int = (1 << 62),
                     skip_last_row: bool = True) -> int:
        done = C.c_uint64()
        self._check(lib().mpc_compress_npy(self._h, path.encode(), first_row, n_rows,
                                           1 if skip_last_row else 0, C.byref(done)))
        return int(done.value)

    def compress_gpgpusim_log(self, path: str):
        """Stream a GPGPU-Sim ``.log`` trace (reference ``LoaderGPGPU.cpp`` + the driver's
        GLOBAL_ACC_R / GLOBAL_ACC_W filter, ``main.cpp:222-224``); returns
        (requests read, lines evaluated)."""
        req, done = C.c_uint64(), C.c_uint64()
        self._check(lib().mpc_compress_gpgpusim_log(self._h, path.encode(), C.byref(req), C.byref(done)))
        return int(req.value), int(done.value)

    def sync(self) -> None:
        self._check(lib().mpc_sync(self._h))

    # -- statistics -----------------------------------------------------------
    def stats_vector(self) -> np.ndarray:
        v = np.zeros(self.stats_len, dtype=np.uint64)
        self._check(lib().mpc_stats_get(self._h, v.ctypes.data, self.stats_len))
        return v

    def stats_raw_len(self) -> int:
        n = C.c_uint64()
        self._check(lib().mpc_stats_raw_len(self._h, C.byref(n)))
        return int(n.value)

    def stats_copy_raw_device(self, d_dst: int, stream: int = 0) -> None:
        """Asynchronous device-to-device copy of the raw uint64 accumulators into ``d_dst``
        (``stats_raw_len()`` words) on ``stream``: the operand of a device-side all-reduce."""
        self._check(lib().mpc_stats_copy_raw_device(self._h, d_dst, stream or None))

    def stats_from_raw(self, raw: np.ndarray) -> np.ndarray:
        """Statistics vector (ABI layout) of a raw accumulator array, e.g. an all-reduced one."""
        raw = np.ascontiguousarray(raw, dtype=np.uint64)
        v = np.zeros(self.stats_len, dtype=np.uint64)
        self._check(lib().mpc_stats_from_raw(self._h, raw.ctypes.data, raw.size, v.ctypes.data, v.size))
        return v

    def stats_merge(self, vec: np.ndarray) -> None:
        vec = np.ascontiguousarray(vec, dtype=np.uint64)
        self._check(lib().mpc_stats_merge(self._h, vec.ctypes.data, len(vec)))

    def stats_set(self, vec: np.ndarray) -> None:
        vec = np.ascontiguousarray(vec, dtype=np.uint64)
        self._check(lib().mpc_stats_set(self._h, vec.ctypes.data, len(vec)))

    def reset(self) -> None:
        self._check(lib().mpc_stats_reset(self._h))


class VPC(_Evaluator):
    """``comp::VPC(configPath)`` (reference ``VPC.h:244-249``)."""

    def __init__(self, config, device: int = -1):
        super().__init__()
        if isinstance(config, dict):
            rc = lib().mpc_create_vpc_from_string(json.dumps(config).encode(), device, C.byref(self._h))
        else:
            rc = lib().mpc_create_vpc(str(config).encode(), device, C.byref(self._h))
        if rc != 0:
            raise MpcError(rc, (lib().mpc_last_error(None) or b"").decode())
        self._finish()
        self.num_modules = self.info.num_modules
        self.hist_bins = self.info.hist_bins
        self.kernel_path = self.info.kernel_path
        self.path_reason = (lib().mpc_path_reason(self._h) or b"").decode()   # why the generic kernel, if it is
        # "unrolled" | "unrolled, general layout" | "unrolled, compiled at creation[ (from the cache)]" | "run-time loop" | "generic"
        self.kernel_form = (lib().mpc_kernel_form(self._h) or b"").decode()

    def result(self) -> Dict:
        """``VPCResult`` (reference ``VPC.h:36-76``) derived from the integer vector."""
        return vpc_result_from_vector(self.stats_vector(), self.num_modules, self.hist_bins, self.line_size)


class BDI(_Evaluator):
    """``comp::BDI(lineSize)`` (reference ``BDI.h:90-107``)."""

    def __init__(self, line_size: int, device: int = -1):
        super().__init__()
        rc = lib().mpc_create_bdi(line_size, device, C.byref(self._h))
        if rc != 0:
            raise MpcError(rc, (lib().mpc_last_error(None) or b"").decode())
        self._finish()
        self.kernel_path = self.info.kernel_path

    def result(self) -> Dict:
        v = self.stats_vector()
        return {"lines": int(v[0]), "original_bits": int(v[1]), "compressed_bits": int(v[2]),
                "comp_ratio": (float(v[1]) / float(v[2])) if v[2] else 0.0,
                "counts": [int(x) for x in v[3:12]]}


class FPC(_Evaluator):
    """``comp::FPC(lineSize)`` (reference ``FPC.h:91-103``); per-line ``selected`` is always 0."""

    def __init__(self, line_size: int, device: int = -1):
        super().__init__()
        rc = lib().mpc_create_fpc(line_size, device, C.byref(self._h))
        if rc != 0:
            raise MpcError(rc, (lib().mpc_last_error(None) or b"").decode())
        self._finish()
        self.kernel_path = self.info.kernel_path

    def result(self) -> Dict:
        v = self.stats_vector()
        return {"lines": int(v[0]), "original_bits": int(v[1]), "compressed_bits": int(v[2]),
                "comp_ratio": (float(v[1]) / float(v[2])) if v[2] else 0.0,
                "total_words": int(v[3:11].sum()), "counts": [int(x) for x in v[3:11]]}


class BPC(_Evaluator):
    """``comp::BPC(lineSize)`` (reference ``BPC.h:90-107``); per-line ``selected`` is always 0."""

    def __init__(self, line_size: int, device: int = -1):
        super().__init__()
        rc = lib().mpc_create_bpc(line_size, device, C.byref(self._h))
        if rc != 0:
            raise MpcError(rc, (lib().mpc_last_error(None) or b"").decode())
        self._finish()
        self.kernel_path = self.info.kernel_path

    def result(self) -> Dict:
        v = self.stats_vector()
        return {"lines": int(v[0]), "original_bits": int(v[1]), "compressed_bits": int(v[2]),
                "comp_ratio": (float(v[1]) / float(v[2])) if v[2] else 0.0,
                "total_words": int(v[3]), "counts": [int(x) for x in v[4:11]]}


def vpc_result_from_vector(v: np.ndarray, M: int, bins: int, L: int) -> Dict:
    K = M + 1
    out = {"lines": int(v[0]), "original_bits": int(v[1]), "compressed_bits": int(v[2]),
           "comp_ratio": (float(v[1]) / float(v[2])) if v[2] else 0.0, "clusters": {}}
    for k in range(K):
        cnt, ob, cb, rl, sr, sr2 = (int(x) for x in v[3 + 6 * k: 3 + 6 * k + 6])
        out["clusters"][k - 1] = {
            "count": cnt, "original_bits": ob, "compressed_bits": cb,
            "comp_ratio": (float(ob) / float(cb)) if cb else 0.0,
            "residue_lines": rl, "sum_r": sr, "sum_r2": sr2,
            # VPC.h:62-76: mean over lines of (sum over bytes / L)
            "mae": (float(sr) / float(L)) / float(rl) if rl else 0.0,
            "mse": (float(sr2) / float(L)) / float(rl) if rl else 0.0,
            "hist": v[3 + 6 * K + k * bins: 3 + 6 * K + (k + 1) * bins].copy(),
        }
    return out


def synth_fill(d_ptr: int, n_lines: int, line_size: int, kind: str, first_line: int = 0,
               seed: int = 12345, stream: int = 0) -> None:
    rc = lib().mpc_synth_fill(d_ptr, n_lines, line_size, SYNTH_KINDS[kind], first_line, seed, stream or None)
    if rc != 0:
        raise MpcError(rc, "mpc_synth_fill failed")


def read_bandwidth_probe(d_ptr: int, nbytes: int, stream: int = 0) -> None:
    rc = lib().mpc_read_bandwidth_probe(d_ptr, nbytes, stream or None)
    if rc != 0:
        raise MpcError(rc, "mpc_read_bandwidth_probe failed")
