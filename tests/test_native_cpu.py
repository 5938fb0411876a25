"""CPU-only native checks: the host mirror of trace::LoaderNPY against numpy (incl. the
reference's dropped last row), and AddressSanitizer / UBSan runs of the configuration
reader and of the oracle (sanitizers are not available on the GPU pool, so here)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "cal_22-mpc_amd", "host")
CSRC = os.path.join(ROOT, "cal_22-mpc_amd", "csrc")
NATIVE = os.path.join(ROOT, "tests", "native")
SAN = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


def _build(tmp, name, cmd):
    out = os.path.join(tmp, name)
    subprocess.run(cmd + ["-o", out], check=True, capture_output=True, text=True)
    return out


def _fnv(data: bytes) -> int:
    h = 1469598103934665603
    for b in data:
        h = ((h ^ b) * 1099511628211) & ((1 << 64) - 1)
    return h


@pytest.fixture(scope="module")
def loader_probe(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("native"))
    return _build(tmp, "loader_probe", ["g++", "-std=c++17", *SAN, "-I", HOST, os.path.join(NATIVE, "loader_probe.cpp"),
                                        os.path.join(HOST, "LoaderNPY.cpp"), os.path.join(HOST, "LoaderGPGPU.cpp"),
                                        os.path.join(HOST, "LoaderAPSim.cpp"), os.path.join(HOST, "utils.cpp")])


@pytest.mark.parametrize("n,L", [(1, 64), (2, 64), (1000, 32), (777, 128)])
def test_loader_npy_matches_numpy(loader_probe, traces, tmp_path, n, L):
    lines = traces.structured(n, L, seed=n)
    p = traces.save_npy(str(tmp_path / "t.npy"), lines)
    want = _fnv(lines[:-1].tobytes())      # every row but the last (LoaderNPY.cpp:28-32 + main.cpp:240)
    for mode in (["line"], ["batch", "100"], ["batch", "1"]):
        r = subprocess.run([loader_probe, p] + mode, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        out = r.stdout.strip().split("\n")
        assert out[0] == f"lines {n} line_size {L}"
        assert out[1] == f"delivered {n - 1} hash {want}"


def test_loader_npy_rejects_bad_files(loader_probe, tmp_path):
    np.save(str(tmp_path / "f.npy"), np.zeros((4, 64), dtype=np.float32))
    np.save(str(tmp_path / "d3.npy"), np.zeros((4, 4, 4), dtype=np.uint8))
    np.save(str(tmp_path / "fo.npy"), np.asfortranarray(np.zeros((4, 64), dtype=np.uint8)))
    (tmp_path / "junk.npy").write_bytes(b"not an npy file at all")
    for name in ("f.npy", "d3.npy", "fo.npy", "junk.npy", "missing.npy"):
        r = subprocess.run([loader_probe, str(tmp_path / name), "line"], capture_output=True, text=True)
        assert r.returncode == 1 and "Invalid File!" in r.stdout, (name, r.stdout, r.stderr)


@pytest.mark.parametrize("n,L", [(1, 64), (300, 32), (777, 128)])
def test_loader_gpgpusim_log_matches_restatement(loader_probe, traces, tmp_path, n, L):
    """The C++ LoaderGPGPU (per-record interface + the driver's request-type filter, and the
    additive GetBatch) against oracle/gpgpusim_log.py on synthetic .log traces: every request
    type, an incomplete trailing record, an incomplete trailing header."""
    sys.path.insert(0, ROOT)
    from oracle import gpgpusim_log as G
    lines = traces.structured(n, L, seed=n + 1)
    types = np.random.default_rng(n).integers(0, 9, n)
    one = open(traces.write_gpgpusim_log(str(tmp_path / "one.log"), lines[:1]), "rb").read()[1 + 7 * 17:]
    for k, tail in enumerate((b"", one[:-1], one[:17])):     # complete; last data byte missing; header cut off
        p = traces.write_gpgpusim_log(str(tmp_path / f"t{k}.log"), lines, types, tail=tail)
        kept = G.evaluated_lines(p)
        assert (kept == lines[(types == 0) | (types == 4)]).all() and G.line_size(p) == L
        want = _fnv(kept.tobytes())
        for mode in (["line"], ["batch", "100"], ["batch", "1"]):
            r = subprocess.run([loader_probe, p] + mode, capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
            out = r.stdout.strip().split("\n")
            assert out[0] == f"lines {n} line_size {L}"           # GetNumLines counts every complete record
            assert out[1] == f"delivered {len(kept)} hash {want}"


def test_loader_gpgpusim_log_rejects_bad_files(loader_probe, traces, tmp_path):
    lines = traces.structured(10, 64)
    good = open(traces.write_gpgpusim_log(str(tmp_path / "g.log"), lines), "rb").read()
    (tmp_path / "keys.log").write_bytes(bytes([16]) + good[1:])          # wrong key count
    (tmp_path / "short.log").write_bytes(good[:50])                      # header cut off
    for name in ("keys.log", "short.log"):
        r = subprocess.run([loader_probe, str(tmp_path / name), "line"], capture_output=True, text=True)
        assert r.returncode == 1 and "header of the GPGPU-sim trace file is not valid" in r.stdout, (name, r.stdout)
    r = subprocess.run([loader_probe, str(tmp_path / "missing.log"), "line"], capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to open a file" in r.stdout
    # requests of two sizes among the evaluated types: refused by the batch interface (documented deviation)
    a = open(traces.write_gpgpusim_log(str(tmp_path / "a.log"), lines), "rb").read()
    b = open(traces.write_gpgpusim_log(str(tmp_path / "b.log"), traces.structured(4, 32)), "rb").read()
    (tmp_path / "mixed.log").write_bytes(a + b[1 + 7 * 17:])
    r = subprocess.run([loader_probe, str(tmp_path / "mixed.log"), "batch", "100"], capture_output=True, text=True)
    assert r.returncode == 1 and "mixes request sizes" in r.stdout


def test_config_reader_under_sanitizers(tmp_path, configs):
    probe = _build(str(tmp_path), "config_probe", ["g++", "-std=c++17", *SAN, "-I", CSRC,
                                                   os.path.join(NATIVE, "config_probe.cpp")])
    files = []
    for L in (32, 64, 128):
        files.append(configs.write_config(configs.probe_config(L), str(tmp_path / f"probe{L}.json")))
    good = json.dumps(configs.probe_config(64))
    rng = np.random.default_rng(3)
    # truncations, byte flips and structural damage: must produce error codes, never a sanitizer report
    for i in range(120):
        if i % 3 == 0:
            bad = good[: int(rng.integers(1, len(good)))]
        elif i % 3 == 1:
            b = bytearray(good.encode())
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(32, 127))
            bad = b.decode(errors="replace")
        else:
            cfg = configs.probe_config(64)
            m = cfg["modules"][str(int(rng.integers(2, 6)))]["submodules"]
            key = ["ScanModule", "ResidueModule", "XORModule"][int(rng.integers(0, 3))]
            if key == "ScanModule":
                m[key]["TableSize"] = int(rng.integers(-5, 2000))
                m[key]["Rows"] = m[key]["Rows"][: int(rng.integers(0, 512))]
            elif key == "ResidueModule":
                m[key]["PredictorModule"]["RootIndex"] = int(rng.integers(-3, 200))
                m[key]["PredictorModule"].pop("BaseIndexTable", None)
            else:
                m[key] = "x"
            bad = json.dumps(cfg)
        p = tmp_path / f"bad{i}.json"
        p.write_text(bad)
        files.append(str(p))
    r = subprocess.run([probe] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = r.stdout.strip().split("\n")
    assert len(out) == len(files)
    assert all(": rc 0 " in line and "path fast" in line for line in out[:3])
    assert sum(": rc -" in line for line in out[3:]) > 60


def test_oracle_under_sanitizers(tmp_path):
    probe = _build(str(tmp_path), "oracle_probe", ["gcc", "-std=c11", *SAN, "-I", os.path.join(ROOT, "oracle"),
                                                   os.path.join(NATIVE, "oracle_probe.c"),
                                                   os.path.join(ROOT, "oracle", "mpc_oracle.c"), "-lm"])
    r = subprocess.run([probe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr[-3000:]


def test_double_text_matches_fmt_layout(tmp_path):
    """mpctext::num lays doubles out like {fmt}'s "{}" (what the reference's CSV writers use): shortest
    round-trip digits, fixed notation for decimal exponents -4 .. 15, d.ddde[+-]XX otherwise.  The expected
    strings are {fmt}'s documented behaviour; Python's repr follows the same rule apart from a trailing
    ".0", which makes it a second witness on random values."""
    probe = _build(str(tmp_path), "num_probe", ["g++", "-std=c++17", *SAN, "-I", HOST, os.path.join(NATIVE, "num_probe.cpp"),
                                                os.path.join(HOST, "utils.cpp")])
    known = {0.0001: "0.0001", 0.0005: "0.0005", 1e5: "100000", 3e5: "300000", 1e16: "1e+16", 1e-5: "1e-05",
             1e15: "1000000000000000", 170.66666666666666: "170.66666666666666", 0.9941747572815534: "0.9941747572815534",
             3.0: "3", 0.0: "0", 1.5e-7: "1.5e-07", 123456789012345680.0: "1.2345678901234568e+17", 0.00012345: "0.00012345",
             64.0 / (64 * 2000): "0.0005", 2.5: "2.5", 1e100: "1e+100", 1234.5e-9: "1.2345e-06", -0.25: "-0.25"}
    rng = np.random.default_rng(11)
    rand = [float(x) for x in np.concatenate([rng.random(300) * 10.0 ** rng.integers(-9, 18, 300),
                                              rng.integers(1, 1 << 40, 200) / rng.integers(1, 1 << 20, 200)])]

    def fmt_like(x):
        r = repr(x)
        if "e" in r:
            m, e = r.split("e")
            return (m[:-2] if m.endswith(".0") else m) + "e" + e
        return r[:-2] if r.endswith(".0") else r
    vals = list(known) + rand
    r = subprocess.run([probe] + [float(v).hex() for v in vals], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.strip().split("\n")
    assert len(out) == len(vals)
    for v, got in zip(vals, out):
        assert got == known.get(v, fmt_like(v)), (v, got)


@pytest.mark.parametrize("write_trace,line_size", [(False, 32), (True, 32), (False, 64)])
def test_loader_apsim_txt_matches_restatement(loader_probe, traces, tmp_path, write_trace, line_size):
    """trace::apsim::LoaderGPGPU (host mirror) on synthetic APSim .txt traffic files against the Python
    restatement of the reference's reader (oracle/apsim_txt.py): handshake filter, channel order, 32-byte
    beats and 64-byte two-beat lines, a last row without a newline dropped, read and write headers."""
    sys.path.insert(0, ROOT)
    from oracle import apsim_txt as A
    beats = np.concatenate([traces.structured(700, 32, seed=8), traces.random_u32(200, 32), traces.zeros(30, 32)])
    beats = beats[np.random.default_rng(1).permutation(len(beats))]
    for final_newline in (True, False):
        p = traces.write_apsim_txt(str(tmp_path / f"t{int(final_newline)}.txt"), beats, write_trace=write_trace,
                                   final_newline=final_newline)
        want = A.lines(p, line_size)
        if line_size == 32:
            # every beat but (without the final newline) those of the last row comes back, in order
            assert len(want) <= len(beats) and (want == beats[: len(want)]).all()
            assert (len(want) == len(beats)) == final_newline
        for mode in (["line"], ["batch", "7", str(line_size)], ["batch", "100000", str(line_size)]):
            args = mode if mode[0] == "batch" else ["line", "0", str(line_size)]
            r = subprocess.run([loader_probe, p] + args, capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr[-3000:]
            out = r.stdout.strip().split("\n")
            assert out[0] == f"lines {len(want)} line_size {line_size}"
            assert out[1] == f"delivered {len(want)} hash {_fnv(want.tobytes())}"


def test_loader_apsim_txt_rejects_bad_files(loader_probe, traces, tmp_path):
    (tmp_path / "nohdr.txt").write_text("Time,clk,a,b\n1,1\n")
    (tmp_path / "empty.txt").write_text("")
    good = traces.write_apsim_txt(str(tmp_path / "g.txt"), traces.random_u32(5, 32))
    text = open(good).read().split("\n")
    (tmp_path / "short.txt").write_text("\n".join(text[:2]) + "\n1,1,1,0\n")
    (tmp_path / "badhex.txt").write_text(text[0] + "\n" + text[1].replace(text[1].split(",")[6], "zz" * 32) + "\n")
    for name, msg in (("nohdr.txt", "header of the GPU traffic file"), ("empty.txt", "header of the GPU traffic file"),
                      ("missing.txt", "Failed to open"), ("short.txt", "row of the GPU traffic file"),
                      ("badhex.txt", "row of the GPU traffic file")):
        r = subprocess.run([loader_probe, str(tmp_path / name), "line"], capture_output=True, text=True)
        assert r.returncode == 1 and msg in r.stdout, (name, r.stdout, r.stderr[-500:])


def test_reference_style_subclass_compiles_against_compressor_h(traces, tmp_path):
    """A compressor that overrides only CompressLine() -- the reference's two-method interface
    (src/compressor/Compressor.h:18-33), like its CPACK / SC2 classes -- compiles against
    cal_22-mpc_amd/host/Compressor.h unchanged, and the defaulted additive CompressBatch() / CompressFile() /
    GetLineSize() give what the reference driver's per-line loop gives (.npy: all rows but the last; .log: the
    GLOBAL_ACC_R / GLOBAL_ACC_W requests).  ASan / UBSan build, no device library."""
    probe = _build(str(tmp_path), "subclass_probe",
                   ["g++", "-std=c++17", *SAN, "-Wall", "-Werror", "-I", HOST, os.path.join(NATIVE, "subclass_probe.cpp"),
                    os.path.join(HOST, "Compressor.cpp"), os.path.join(HOST, "CompResult.cpp"), os.path.join(HOST, "LoaderNPY.cpp"),
                    os.path.join(HOST, "LoaderGPGPU.cpp"), os.path.join(HOST, "LoaderAPSim.cpp"), os.path.join(HOST, "utils.cpp")])
    L = 64
    lines = traces.structured(1234, L, seed=9)
    npy = traces.save_npy(str(tmp_path / "t.npy"), lines)
    types = np.random.default_rng(3).integers(0, 9, len(lines))
    log = traces.write_gpgpusim_log(str(tmp_path / "t.log"), lines, types)
    for path, kept in ((npy, lines[:-1]), (log, lines[(types == 0) | (types == 4)])):
        r = subprocess.run([probe, path, str(L)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        want = f"{len(kept)} {len(kept) * L * 8} {int((kept != 0).sum()) * 8 + len(kept)}"
        assert r.stdout.strip().split("\n") == [f"line {want}", f"batch {want}", f"file {want}"], r.stdout
