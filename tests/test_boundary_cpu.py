"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/mpc_hip.h declares; the native configuration
parser agrees with the oracle's reading of the same JSON; errors are reported as
codes, not exits.  No kernel runs here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT, pkg


@pytest.fixture(scope="module")
def mpc():
    pkg("build").build_lib()
    return pkg()


def test_header_symbols_exported(mpc):
    with open(os.path.join(ROOT, "include", "mpc_hip.h")) as f:
        hdr = f.read()
    declared = set(re.findall(r"\b(mpc_[a-z_]+)\s*\(", hdr))
    declared.discard("mpc_handle")
    assert declared, "no declarations found"
    lib = C.CDLL(mpc.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in mpc_hip.h but not exported"
    assert declared == set(mpc.EXPORTED_SYMBOLS)


def test_config_describe_matches_oracle_parse(mpc, configs, oracle):
    for L in (32, 64, 128):
        cfg = configs.probe_config(L)
        d = mpc.describe_config(cfg)
        assert d["rc"] == 0 and d["path"] == "fast" and d["sequence"] == "unrolled", d
        oc = oracle.config_from_json(cfg)
        assert d["L"] == oc.line_size and d["M"] == oc.num_modules
        assert d["enc_bits"] == [oc.enc_bits[k] for k in range(oc.num_modules + 1)]
        assert d["hist_bins"] == oracle.hist_bins(oc)
        for i, m in enumerate(d["modules"]):
            om = oc.modules[i]
            assert m["kind"] == om.kind
            if om.kind == 2:
                assert (m["pred_kind"], m["root"], m["cx"], m["table_size"]) == \
                    (om.pred_kind, om.root, om.consecutive_xor, om.table_size)
        # weight 1.0 -> shift 0, weight 0.5 -> shift -1
        assert d["modules"][5]["shifts"][1:4] == [-1, 0, -1]


def test_custom_encoding_bits_and_generic_classification(mpc, configs):
    cfg = configs.probe_config(64, encoding_bits=[2, 1, 3, 4, 4, 5, 5])
    d = mpc.describe_config(cfg)
    assert d["enc_bits"] == [2, 1, 3, 4, 4, 5, 5] and d["hist_bins"] == 512 + 5 + 1
    # a root of 1..15 / a truncated plane-major table -> the unrolled kernels' general-layout twins
    cfg = configs.make_config(64, [{"name": "AllZero"}, configs.one_base(64, root=7)])
    d = mpc.describe_config(cfg)
    assert d["rc"] == 0 and d["path"] == "fast" and d["sequence"] == "unrolled" and d["general_layout"] == "yes"
    assert mpc.describe_config(configs.probe_config(64))["general_layout"] == "no"
    ts = 6 * 64
    cfg = configs.make_config(64, [{"name": "AllZero"}, configs.one_base(64, scan={"TableSize": ts, "Rows": [i // 64 for i in range(ts)],
                                                                                "Cols": [i % 64 for i in range(ts)]})])
    d = mpc.describe_config(cfg)
    assert d["rc"] == 0 and d["path"] == "fast" and d["sequence"] == "unrolled" and d["general_layout"] == "yes"
    # a root above 15: no built-in kernel takes it, the general-layout kernel is compiled at creation with the roots as constants
    cfg = configs.make_config(64, [{"name": "AllZero"}, configs.one_base(64, root=16)])
    d = mpc.describe_config(cfg)
    assert d["rc"] == 0 and d["path"] == "fast" and d["sequence"] == "unrolled" and d["compiled"] == "at creation" and d["general_layout"] == "yes"
    # a table without a complete first row -> still the fast kernel, through its run-time module loop
    for ts, general in ((300, True), (16, True), (12, False)):      # cut inside a bit plane; one complete row; less than a row
        cfg = configs.make_config(64, [{"name": "AllZero"}, configs.one_base(64, scan={"TableSize": ts, "Rows": [i // 64 for i in range(ts)],
                                                                                    "Cols": [i % 64 for i in range(ts)]})])
        d = mpc.describe_config(cfg)
        assert d["rc"] == 0 and d["path"] == "fast" and d["sequence"] == ("unrolled" if general else "run-time loop"), (ts, d)
        assert d["general_layout"] == ("yes" if general else "no"), (ts, d)
    # permuted scan table -> generic kernel
    scan = configs.plane_major_scan(64)
    scan["Rows"][0], scan["Rows"][100] = scan["Rows"][100], scan["Rows"][0]
    cfg = configs.make_config(64, [{"name": "AllZero"}, configs.one_base(64, scan=scan)])
    assert mpc.describe_config(cfg)["path"] == "generic"
    # base table outside the own/previous dword -> compiled at creation with the table as constants (a byte gather); the generic
    # kernel only when that is switched off, or when the table comes with another layout the built-in kernels lack
    base = [0] * 64
    cfg = configs.make_config(64, [{"name": "AllZero"}, configs.diff_base(64, base, [0] * 64)])
    d = mpc.describe_config(cfg)
    assert d["path"] == "fast" and d["sequence"] == "unrolled" and d["compiled"] == "at creation", d
    assert mpc.jit_compile_check(cfg) > 10000
    os.environ["MPC_JIT"] = "0"
    try:
        d = mpc.describe_config(cfg)
        assert d["path"] == "generic" and "not windowed" in d["why_generic"], d
    finally:
        del os.environ["MPC_JIT"]
    bm = {"TableSize": 512, "Rows": [i % 8 for i in range(512)], "Cols": [i // 8 for i in range(512)]}
    cfg = configs.make_config(64, [{"name": "AllZero"}, configs.diff_base(64, base, [0] * 64, 0, True, bm)])
    assert mpc.describe_config(cfg)["path"] == "generic"


def test_invalid_configs_are_error_codes(mpc, configs):
    assert mpc.describe_config("{ not json")["rc"] == -74
    cfg = configs.probe_config(64)
    cfg["modules"]["3"]["name"] = "Bogus"
    d = mpc.describe_config(cfg)
    assert d["rc"] == -22 and "not a valid compression module" in d["error"]
    cfg = configs.probe_config(64)
    cfg["modules"]["0"], cfg["modules"]["2"] = cfg["modules"]["2"], cfg["modules"]["0"]
    assert mpc.describe_config(cfg)["rc"] == -22          # module 0 must be AllZero
    cfg = configs.make_config(64, [{"name": "AllZero"}, configs.consecutive_base(64, root=2)])
    assert mpc.describe_config(cfg)["rc"] == -22          # inputLine[-1] in the reference
    cfg = configs.probe_config(64)
    cfg["modules"]["2"]["submodules"]["ResidueModule"]["PredictorModule"]["LineSize"] = 32
    assert mpc.describe_config(cfg)["rc"] == -22


def test_json_reader_jsoncpp_conversions(mpc, configs):
    # reals truncate to int, ints convert to bool, comments are accepted
    cfg = configs.probe_config(32)
    text = json.dumps(cfg).replace('"lineSize": 32', '"lineSize": 32.0 /* c */')
    text = text.replace('"consecutiveXOR": true', '"consecutiveXOR": 1')
    d = mpc.describe_config("// leading comment\n" + text)
    assert d["rc"] == 0 and d["L"] == 32 and d["modules"][2]["cx"] == 1


def test_create_without_device_fails_loudly(mpc, configs):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(mpc.MpcError) as e:
        mpc.VPC(configs.probe_config(64))
    assert e.value.code == -19 and "no CPU fallback" in str(e.value)
    with pytest.raises(mpc.MpcError):
        mpc.BDI(64)


def test_npy_shape_probe(mpc, traces, tmp_path):
    p = str(tmp_path / "t.npy")
    traces.save_npy(p, traces.random_u32(37, 64))
    r, c = C.c_uint64(), C.c_uint64()
    assert mpc.lib().mpc_npy_shape(p.encode(), C.byref(r), C.byref(c)) == 0
    assert (r.value, c.value) == (37, 64)
    np.save(str(tmp_path / "f.npy"), np.zeros((4, 64), dtype=np.float32))
    assert mpc.lib().mpc_npy_shape(str(tmp_path / "f.npy").encode(), C.byref(r), C.byref(c)) == -22


def test_gpgpusim_log_probe_needs_no_device(mpc, traces, tmp_path):
    lines = traces.structured(20, 128)
    p = traces.write_gpgpusim_log(str(tmp_path / "t.log"), lines)
    assert mpc.gpgpusim_log_line_size(p) == 128
    (tmp_path / "bad.log").write_bytes(b"\x10" + bytes(200))
    with pytest.raises(mpc.MpcError) as e:
        mpc.gpgpusim_log_line_size(str(tmp_path / "bad.log"))
    assert e.value.code == -74 and "header of the GPGPU-sim trace file is not valid" in str(e.value)
    with pytest.raises(mpc.MpcError) as e:
        mpc.gpgpusim_log_line_size(str(tmp_path / "missing.log"))
    assert e.value.code == -2


def test_element_configs_are_fast_path(mpc, configs, oracle):
    """The authoring helper's configurations (1/2/4/8-byte elements, every line size) parse, map to
    the fast kernel, and equal the probe configurations where those exist."""
    for L in (32, 64, 128):
        for e in (1, 2, 4, 8):
            cfg = configs.element_config(L, e)
            d = mpc.describe_config(cfg)
            assert d["rc"] == 0 and d["path"] == "fast", (L, e, d)
            oracle.VpcOracle(cfg)
        assert configs.element_config(L, 4) == configs.probe_config(L)
        assert configs.element_config(L, 8) == configs.probe_config_u64(L)



def test_npy_header_malformed_is_an_error_code(mpc, tmp_path):
    """Hand-made .npy headers that end inside a key's value must come back as an error code from the C ABI,
    never as a C++ exception escaping an extern "C" function (std::terminate would take the host down)."""
    import struct

    def write(name, header: bytes):
        p = tmp_path / name
        p.write_bytes(b"\x93NUMPY\x01\x00" + struct.pack("<H", len(header)) + header)
        return str(p)
    rows, cols = C.c_uint64(), C.c_uint64()
    bad = [b"{'descr': '|u1', 'fortran_order':", b"{'descr': '|u1', 'fortran_order':     ", b"{'descr': '|u1",
           b"{'descr': '|u1', 'fortran_order': False, 'shape': (3, 64", b"{}", b"",
           b"{'descr': '|u1', 'fortran_order': True, 'shape': (3, 64), }"]
    for i, h in enumerate(bad):
        rc = mpc.lib().mpc_npy_shape(write(f"bad{i}.npy", h).encode(), C.byref(rows), C.byref(cols))
        assert rc < 0, (i, h, rc)
    ok = write("ok.npy", b"{'descr': '|u1', 'fortran_order': False, 'shape': (3, 64), }" + b" " * 10 + b"\n")
    assert mpc.lib().mpc_npy_shape(ok.encode(), C.byref(rows), C.byref(cols)) == 0 and (rows.value, cols.value) == (3, 64)


def test_datatype_model_configs_are_fast_path(mpc, configs, oracle):
    """The per-datatype models of the authoring tool (the five prediction models of the paper's overview
    figure and each data type alone) parse, are accepted by the oracle and map to the fast kernel."""
    for L in (32, 64, 128):
        for dt in configs.DTYPE_BYTES:
            cfg = configs.datatype_config(L, dt)
            d = mpc.describe_config(cfg)
            assert d["rc"] == 0 and d["path"] == "fast", (L, dt, d)
            oracle.VpcOracle(cfg)
        d = mpc.describe_config(configs.mpc_config(L))
        assert d["rc"] == 0 and d["path"] == "fast" and d["M"] == 7, (L, d)
        oracle.VpcOracle(configs.mpc_config(L))


def test_short_scan_tables_pin_the_scan_order(mpc, configs):
    """Entry i of a plane-major scan table is (row 0, column i) for i < L, of a byte-major one (row i % 8, column
    i / 8): only a table of at most one entry is both (and is taken as plane-major, the test that runs first).
    Truncated byte-major tables of 2 .. 8 entries -- a single column -- must be classified byte-major, the
    plane-major ones of the same sizes plane-major; a table that is neither goes to the generic kernel."""
    L = 64
    for ts in range(0, 10):
        bm = {"TableSize": ts, "Rows": [i % 8 for i in range(ts)], "Cols": [i // 8 for i in range(ts)]}
        pm = {"TableSize": ts, "Rows": [i // L for i in range(ts)], "Cols": [i % L for i in range(ts)]}
        for want, sc in (("byte-major" if ts >= 2 else "plane-major", bm), ("plane-major", pm)):
            mods = [{"name": "AllZero"}, configs.one_base(L, 0, True, sc), configs.consecutive_base(L, 0, False, sc)]
            d = mpc.describe_config(configs.make_config(L, mods))
            assert d["rc"] == 0 and d["path"] == "fast" and d["sequence"] == "run-time loop", (ts, d)
            assert d["scan_order"] == want, (ts, want, d["scan_order"])
    # the two orders in one configuration: no fast form
    ts = 5
    bm = {"TableSize": ts, "Rows": [i % 8 for i in range(ts)], "Cols": [i // 8 for i in range(ts)]}
    pm = {"TableSize": ts, "Rows": [0] * ts, "Cols": list(range(ts))}
    d = mpc.describe_config(configs.make_config(L, [{"name": "AllZero"}, configs.one_base(L, 0, True, bm), configs.one_base(L, 0, False, pm)]))
    assert d["path"] == "generic"
    assert mpc.describe_config(configs.probe_config(L))["scan_order"] == "plane-major"


def test_run_time_compilation_of_a_module_sequence_builds_for_gfx950(mpc, configs):
    """A module sequence without a built-in unrolled instantiation is compiled with hiprtc when a handle is created
    (csrc/mpc_jit.h).  The compilation itself needs no device: the same source the library would hand to hiprtc on a GPU
    box is compiled here for gfx950 (nothing is loaded), at every line size, with and without the general layout."""
    az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}
    for L in (32, 64, 128):
        prev1 = [max(i - 1, 0) for i in range(L)]
        prev4 = [max(i - 4, 0) for i in range(L)]
        w2 = [[1.0, 0.5][i % 2] for i in range(L)]
        w3 = [[2.0, 0.25][i % 2] for i in range(L)]                 # two shifted classes (the WEIGHT2 form)
        seqs = [[az, aws, configs.one_base(L, 0, True), configs.diff_base(L, prev1, [3] * L, 0, False), configs.weight_base(L, prev4, w2, 0, True),
                 configs.one_base(L, 0, False)],
                [az, configs.consecutive_base(L, 0, True), configs.weight_base(L, prev4, w3, 2, False), configs.one_base(L, 9, True)]]
        for mods in seqs:
            cfg = configs.make_config(L, mods)
            d = mpc.describe_config(cfg)
            assert d["path"] == "fast" and d["sequence"] == "unrolled" and d["compiled"] == "at creation", d
            assert mpc.jit_compile_check(cfg) > 10000
    # built in / run-time loop: nothing to compile
    assert mpc.describe_config(configs.probe_config(64))["compiled"] == "built in"
    assert mpc.jit_compile_check(configs.probe_config(64)) == 0
    # a root above 15 (compiled with the roots as constants) and the byte-major order (its own stages) have no built-in kernel at all
    cfg = configs.make_config(64, [az, configs.one_base(64, root=40), configs.consecutive_base(64, 0, True)])
    assert mpc.describe_config(cfg)["compiled"] == "at creation" and mpc.jit_compile_check(cfg) > 10000
    bm = {"TableSize": 512, "Rows": [i % 8 for i in range(512)], "Cols": [i // 8 for i in range(512)]}
    cfg = configs.make_config(64, [az, configs.one_base(64, 0, True, bm), configs.consecutive_base(64, 0, True, bm)])
    assert mpc.describe_config(cfg)["compiled"] == "at creation" and mpc.jit_compile_check(cfg) > 10000
    # every variant of the generated translation unit, at the other line sizes too: byte-major (truncated), roots above 15 as
    # constants, a different number of bit planes per module, a workgroup smaller than the built-in one (the five models at 128 B)
    for L in (32, 128):
        def pm(ts):
            return {"TableSize": ts, "Rows": [i // L for i in range(ts)], "Cols": [i % L for i in range(ts)]}
        bmL = {"TableSize": 8 * L - 20, "Rows": [i % 8 for i in range(8 * L - 20)], "Cols": [i // 8 for i in range(8 * L - 20)]}
        prev4 = [max(i - 4, 0) for i in range(L)]
        w2 = [[1.0, 0.5][i % 2] for i in range(L)]
        for mods in ([az, configs.one_base(L, 0, True, bmL), configs.weight_base(L, prev4, w2, 0, True, bmL)],
                     [az, configs.one_base(L, L - 1, True), configs.weight_base(L, prev4, w2, 17, True), configs.consecutive_base(L, 0, False)],
                     [az, configs.one_base(L, 2, True, pm(6 * L)), configs.consecutive_base(L, 0, True, pm(8 * L)), configs.weight_base(L, prev4, w2, 0, True, pm(3 * L))]):
            cfg = configs.make_config(L, mods)
            assert mpc.describe_config(cfg)["compiled"] == "at creation", mpc.describe_config(cfg)
            assert mpc.jit_compile_check(cfg) > 10000
    d = mpc.describe_config(configs.mpc_config(128))
    assert d["sequence"] == "unrolled" and d["compiled"] == "at creation" and mpc.jit_compile_check(configs.mpc_config(128)) > 10000
    # nothing to compile: a byte-major table with a non-zero root, more than 12 prediction modules
    cfg = configs.make_config(64, [az, configs.one_base(64, 3, True, bm), configs.consecutive_base(64, 0, True, bm)])
    assert mpc.describe_config(cfg)["sequence"] == "run-time loop" and mpc.jit_compile_check(cfg) == 0
    cfg = configs.make_config(64, [az] + [configs.one_base(64, 0, bool(i & 1)) for i in range(13)])
    assert mpc.describe_config(cfg)["sequence"] == "run-time loop" and mpc.jit_compile_check(cfg) == 0
