"""The `compressor` CLI and `bin/run` driver: flag handling and error behaviour on
CPU; on the GPU box, byte-for-byte CSV / stdout text against the formats of the
reference (src/main.cpp:129-165, VPC.h:78-211, BDI.h:35-84, CompResult.h:37-74)
filled from the oracle's statistics."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, pkg

BIN = os.path.join(ROOT, "bin")


@pytest.fixture(scope="module")
def cli():
    pkg("build").build_all()
    assert os.path.exists(os.path.join(BIN, "compressor"))
    return os.path.join(BIN, "compressor")


def run(cmd, cwd=BIN):
    return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=600)


def fmt_double(x: float) -> str:
    """{fmt}'s "{}" of a double: shortest round-trip, no trailing ".0"."""
    if x != x:
        return "nan"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    r = repr(float(x))
    if "e" in r:
        m, e = r.split("e")
        if m.endswith(".0"):
            m = m[:-2]
        return m + "e" + e
    return r[:-2] if r.endswith(".0") else r


def test_fmt_double_helper():
    assert fmt_double(170.66666666666666) == "170.66666666666666"
    assert fmt_double(3.0) == "3" and fmt_double(0.0) == "0"
    assert fmt_double(0.9941747572815534) == "0.9941747572815534"
    assert fmt_double(1e-05) == "1e-05" and fmt_double(2.5e-07) == "2.5e-07"


def test_help_and_missing_arguments(cli):
    r = run([cli, "-h"])
    assert r.returncode == 0 and "-a, --algorithm arg" in r.stdout and "Default=VPC" in r.stdout
    # no input -> help, exit 0; VPC without -c -> help, exit 0 (reference main.cpp:52-66)
    assert run([cli]).returncode == 0
    r = run([cli, "-a", "VPC", "-i", "x/y.npy"])
    assert r.returncode == 0 and "Usage:" in r.stdout


def test_out_of_scope_algorithms_and_formats(cli, traces, tmp_path):
    d = tmp_path / "ds"
    d.mkdir()
    p = traces.save_npy(str(d / "t.npy"), traces.zeros(8, 64))
    for algo in ("CPACK", "SC2", "PATTERN", "VIEWER"):
        r = run([cli, "-a", algo, "-i", p, "-o", str(tmp_path)])
        assert r.returncode == 1 and "not part of this build" in r.stdout
    r = run([cli, "-a", "BDI", "-i", str(d / "t.txt"), "-o", str(tmp_path)])
    assert r.returncode == 1 and "Failed to open a file" in r.stdout
    r = run([cli, "-a", "BDI", "-i", str(d / "missing.log"), "-o", str(tmp_path)])
    assert r.returncode == 1 and "Failed to open a file" in r.stdout
    r = run([cli, "-a", "BDI", "-i", str(d / "t.bin"), "-o", str(tmp_path)])
    assert r.returncode != 0 and "Unsupported extension" in r.stderr


def test_bad_config_messages(cli, traces, tmp_path):
    import torch
    d = tmp_path / "ds"
    d.mkdir()
    p = traces.save_npy(str(d / "t.npy"), traces.zeros(8, 64))
    r = run([cli, "-a", "VPC", "-i", p, "-c", str(tmp_path / "missing.json"), "-o", str(tmp_path)])
    assert r.returncode == 1 and "is not valid path." in r.stdout
    bad = tmp_path / "bad.json"
    bad.write_text("{ nope")
    r = run([cli, "-a", "VPC", "-i", p, "-c", str(bad), "-o", str(tmp_path)])
    assert r.returncode == 1 and "is not valid json file." in r.stdout
    if not torch.cuda.is_available():
        # no GPU: the product refuses loudly instead of falling back to a CPU path
        r = run([cli, "-a", "BDI", "-i", p, "-o", str(tmp_path)])
        assert r.returncode == 1 and "no CPU fallback" in r.stdout


# ---------------------------------------------------------------------------------
def vpc_expected_rows(o, workload):
    st, M = o.st, o.M
    row = f"{workload},{st.original_bits},{st.compressed_bits},{fmt_double(st.comp_ratio)},"
    for k in range(M + 1):
        row += f"{st.c_original_bits[k]},{st.c_compressed_bits[k]},{fmt_double(st.c_comp_ratio[k])},"
    det = f"{workload},"
    for k in range(M + 1):
        det += f"{fmt_double(st.mae[k])},{fmt_double(st.mse[k])},"
    h = o.hist()
    for k in range(1, M + 1):
        det += "".join(f"{int(h[k][s])}," for s in range(288))
    return row, det


def vpc_headers(M):
    h1 = "workload,total,,," + "".join(f"{i},,," for i in range(-1, M)) + "\n"
    h1 += ",original_size,compressed_size,compression_ratio,count," + "original_size,compressed_size,compression_ratio," * (M + 1) + "\n"
    h2 = "workload," + "".join(f"{i},," for i in range(-1, M)) + "".join(f"{i}," + "," * 287 for i in range(M)) + "\n"
    h2 += "," + "mae,mse," * (M + 1) + "".join(f"{j}," for j in range(288)) * M + "\n"
    return h1, h2


@pytest.mark.gpu
def test_bin_run_vpc_dataset(cli, oracle, configs, traces, tmp_path):
    """bin/run over a dataset directory: split/<x>set.npy -> out/split/, top-level
    .npy -> out/.  Includes BASELINE config 1 (1M all-zero lines -> 999 999 lines
    processed, ratio 170.66666666666666)."""
    ds, out = tmp_path / "data", tmp_path / "out"
    (ds / "splitA").mkdir(parents=True)
    (out / "splitA").mkdir(parents=True)
    cfg = configs.probe_config(64)
    cfg_path = configs.write_config(cfg, str(tmp_path / "probe64.json"))
    zeros = traces.zeros(1_000_000, 64)
    mix = np.concatenate([traces.structured(5000, 64), traces.mixed(3000, 64), traces.word_same(100, 64)])
    traces.save_npy(str(ds / "splitA" / "zeros_testset.npy"), zeros)
    traces.save_npy(str(ds / "splitA" / "mix_trainset.npy"), mix)
    traces.save_npy(str(ds / "top.npy"), mix[:1001])
    # a GPGPU-Sim .log in the split directory: its rows go to out/, not out/<split>/ (bin/run:37-53)
    log_types = np.random.default_rng(6).integers(0, 9, 2000)
    traces.write_gpgpusim_log(str(ds / "splitA" / "kern.log"), mix[:2000], log_types)
    log_lines = mix[:2000][(log_types == 0) | (log_types == 4)]
    r = run([os.path.join(BIN, "run"), "VPC", str(ds), str(out), cfg_path])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    # echo lines + comp.ratio lines, in directory order
    o = oracle.VpcOracle(cfg)
    expect_stdout, sum_rows, det_rows = [], {"splitA": [], "": []}, {"splitA": [], "": []}
    for split, name, data in (("splitA", "mix_trainset", mix[:-1]), ("splitA", "zeros_testset", zeros[:-1]),
                              ("log", "kern", log_lines), ("", "top", mix[:1000])):
        o.reset()
        o.compress(data)                            # .npy: the driver never sees the last row
        # bin/run echoes "$split : name"; for top-level files $split is whatever the
        # directory walk left behind (the last entry: "top.npy"), as in the reference
        label = ("splitA : " if split else "top.npy : ") + name
        expect_stdout += [label, "comp.ratio: " + fmt_double(o.st.comp_ratio)]
        parent = "splitA" if split else "data"
        row, det = vpc_expected_rows(o, f"{parent}_{name}")
        split = "" if split == "log" else split
        sum_rows[split].append(row)
        det_rows[split].append(det)
        if name == "zeros_testset":
            assert o.st.lines == 999_999 and fmt_double(o.st.comp_ratio) == "170.66666666666666"
    assert lines == expect_stdout
    h1, h2 = vpc_headers(6)
    for split in ("splitA", ""):
        d = out / split if split else out
        assert (d / "probe64_results.csv").read_text() == h1 + "".join(r_ + "\n" for r_ in sum_rows[split])
        assert (d / "probe64_results_detail.csv").read_text() == h2 + "".join(r_ + "\n" for r_ in det_rows[split])
    # a second run appends rows without repeating the header
    r = run([os.path.join(BIN, "run"), "VPC", str(ds), str(out), cfg_path])
    assert r.returncode == 0
    assert (out / "probe64_results.csv").read_text() == h1 + "".join(r_ + "\n" for r_ in sum_rows[""]) * 2


@pytest.mark.gpu
def test_cli_bdi(cli, oracle, traces, tmp_path):
    ds = tmp_path / "bench"
    ds.mkdir()
    data = np.concatenate([traces.bdi_stress(2800, 128), traces.pointers_u64(1000, 128)])
    p = traces.save_npy(str(ds / "ptr.npy"), data)
    r = run([cli, "-a", "BDI", "-i", p, "-o", str(tmp_path)])
    assert r.returncode == 0, r.stdout + r.stderr
    o = oracle.BdiOracle(128)
    o.compress(data[:-1])
    assert r.stdout.strip().split("\n")[-1] == "comp.ratio: " + fmt_double(o.st.comp_ratio)
    hdr = ("Workload,Original Size,Compressed Size,Compression Ratio,Zeros,Repeated,B8D1,B8D2,B8D4,B4D1,B4D2,B2D1,"
           "Uncompressed,\n")
    row = f"bench_ptr,{o.st.original_bits},{o.st.compressed_bits},{fmt_double(o.st.comp_ratio)}," + \
        "".join(f"{o.st.counts[i]}," for i in range(9)) + "\n"
    assert (tmp_path / "BDI_results.csv").read_text() == hdr + row
    assert not (tmp_path / "BDI_results_detail.csv").exists()


@pytest.mark.gpu
def test_cli_gpgpusim_log(cli, oracle, configs, traces, tmp_path):
    """GPGPU-Sim .log traces through the CLI: only GLOBAL_ACC_R / GLOBAL_ACC_W requests are
    evaluated (reference main.cpp:222-224), every complete record counts (no dropped last
    row), an incomplete trailing record is ignored.  Expected text from the oracle run on
    oracle/gpgpusim_log.py's reading of the same file."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import gpgpusim_log as G
    ds = tmp_path / "sim"
    ds.mkdir()
    lines = np.concatenate([traces.structured(3000, 64, seed=4), traces.mixed(2000, 64), traces.random_u32(500, 64),
                            traces.zeros(100, 64)])
    lines = lines[np.random.default_rng(2).permutation(len(lines))]
    types = np.random.default_rng(3).integers(0, 9, len(lines))
    one = open(traces.write_gpgpusim_log(str(tmp_path / "one.log"), lines[:1]), "rb").read()[1 + 7 * 17:]
    p = traces.write_gpgpusim_log(str(ds / "kernel7.log"), lines, types, tail=one[:-3])
    kept = G.evaluated_lines(p)
    assert len(kept) == int(((types == 0) | (types == 4)).sum())
    cfg = configs.probe_config(64)
    cfg_path = configs.write_config(cfg, str(tmp_path / "probe64.json"))
    r = run([cli, "-a", "VPC", "-i", p, "-c", cfg_path, "-o", str(tmp_path)])
    assert r.returncode == 0, r.stdout + r.stderr
    o = oracle.VpcOracle(cfg)
    o.compress(kept)
    assert r.stdout.strip().split("\n")[-1] == "comp.ratio: " + fmt_double(o.st.comp_ratio)
    row, det = vpc_expected_rows(o, "sim_kernel7")
    h1, h2 = vpc_headers(6)
    assert (tmp_path / "probe64_results.csv").read_text() == h1 + row + "\n"
    assert (tmp_path / "probe64_results_detail.csv").read_text() == h2 + det + "\n"
    r = run([cli, "-a", "BDI", "-i", p, "-o", str(tmp_path)])
    assert r.returncode == 0, r.stdout + r.stderr
    ob = oracle.BdiOracle(64)
    ob.compress(kept)
    assert r.stdout.strip().split("\n")[-1] == "comp.ratio: " + fmt_double(ob.st.comp_ratio)
    # a trace without any global access: nothing evaluated, CompRatio keeps its initial 0 (CompResult.h:24-27)
    p2 = traces.write_gpgpusim_log(str(ds / "none.log"), lines[:50], np.full(50, 2))
    r = run([cli, "-a", "BDI", "-i", p2, "-o", str(tmp_path)])
    assert r.returncode == 0 and r.stdout.strip().split("\n")[-1] == "comp.ratio: 0"


@pytest.mark.gpu
def test_cli_fpc(cli, oracle, traces, tmp_path):
    ds = tmp_path / "bench"
    ds.mkdir()
    data = np.concatenate([traces.structured(3000, 64), traces.mixed(1000, 64), traces.zeros(40, 64)])
    p = traces.save_npy(str(ds / "app.npy"), data)
    r = run([cli, "-a", "FPC", "-i", p, "-o", str(tmp_path)])
    assert r.returncode == 0, r.stdout + r.stderr
    o = oracle.FpcOracle(64)
    o.compress(data[:-1])
    assert r.stdout.strip().split("\n")[-1] == "comp.ratio: " + fmt_double(o.st.comp_ratio)
    hdr = ("Workload,Original Size,Compressed Size,Compression Ratio,Total Words,Prefix0,Prefix1,Prefix2,Prefix3,Prefix4,"
           "Prefix5,Prefix6,Prefix7,\n")
    row = f"bench_app,{o.st.original_bits},{o.st.compressed_bits},{fmt_double(o.st.comp_ratio)},{o.st.total_words}," + \
        "".join(f"{o.st.counts[i]}," for i in range(8)) + "\n"
    assert (tmp_path / "FPC_results.csv").read_text() == hdr + row
    assert not (tmp_path / "FPC_results_detail.csv").exists()


@pytest.mark.gpu
def test_cli_bpc(cli, oracle, traces, tmp_path):
    ds = tmp_path / "bench"
    ds.mkdir()
    data = np.concatenate([traces.structured(3000, 128), traces.counters_u32(1000, 128), traces.zeros(40, 128)])
    p = traces.save_npy(str(ds / "app.npy"), data)
    r = run([cli, "-a", "BPC", "-i", p, "-o", str(tmp_path)])
    assert r.returncode == 0, r.stdout + r.stderr
    o = oracle.BpcOracle(128)
    o.compress(data[:-1])
    assert r.stdout.strip().split("\n")[-1] == "comp.ratio: " + fmt_double(o.st.comp_ratio)
    hdr = ("Workload,Original Size,Compressed Size,Compression Ratio,Total Words,Pattern0,Pattern1,Pattern2,Pattern3,"
           "Pattern4,Pattern5,Pattern6,\n")
    row = f"bench_app,{o.st.original_bits},{o.st.compressed_bits},{fmt_double(o.st.comp_ratio)},{o.st.total_words}," + \
        "".join(f"{o.st.counts[i]}," for i in range(7)) + "\n"
    assert (tmp_path / "BPC_results.csv").read_text() == hdr + row


@pytest.mark.gpu
def test_cli_generic_kernel_is_announced(cli, configs, traces, tmp_path):
    """A configuration that only the generic (slow) kernel can run is never entered silently: stderr names the reason."""
    d = tmp_path / "ds"
    d.mkdir()
    p = traces.save_npy(str(d / "t.npy"), traces.structured(300, 64))
    perm = [int(x) for x in np.random.default_rng(1).permutation(512)]
    cfg = configs.make_config(64, [{"name": "AllZero"}, configs.one_base(64, 0, True, {"TableSize": 512, "Rows": [q // 64 for q in perm],
                                                                                    "Cols": [q % 64 for q in perm]})])
    r = run([cli, "-a", "VPC", "-i", p, "-c", configs.write_config(cfg, str(tmp_path / "perm.json")), "-o", str(tmp_path)])
    assert r.returncode == 0 and "generic (slow, exact) kernel" in r.stderr and "scan table" in r.stderr
    r = run([cli, "-a", "VPC", "-i", p, "-c", configs.write_config(configs.probe_config(64), str(tmp_path / "probe.json")), "-o", str(tmp_path)])
    assert r.returncode == 0 and "generic" not in r.stderr


@pytest.mark.gpu
def test_cli_module_sequence_compiled_at_creation(cli, oracle, configs, traces, tmp_path):
    """The C++ driver (no Python, no PyTorch in the process) on a configuration whose module sequence has no built-in
    kernel: the library compiles it when `comp::VPC` is constructed (helper process `mpc_jitc`); the ratio the
    reference's `compressLines` prints is the oracle's; with MPC_JIT=0 the run-time loop gives the same text."""
    L = 64
    d = tmp_path / "ds"
    d.mkdir()
    lines = np.concatenate([traces.structured(3000, L, seed=9), traces.mixed(1500, L), traces.random_u32(500, L)])
    p = traces.save_npy(str(d / "t.npy"), lines)
    prev1 = [max(i - 1, 0) for i in range(L)]
    prev4 = [max(i - 4, 0) for i in range(L)]
    cfg = configs.make_config(L, [{"name": "AllZero"}, configs.consecutive_base(L, 0, True), configs.diff_base(L, prev1, [2] * L, 3, False),
                                  configs.weight_base(L, prev4, [[1.0, 0.5][i % 2] for i in range(L)], 0, True), configs.one_base(L, 7, True)])
    cfg_path = configs.write_config(cfg, str(tmp_path / "seq.json"))
    o = oracle.VpcOracle(cfg)
    o.compress(lines[:-1])                               # the reference driver drops the last row
    want = "comp.ratio: " + fmt_double(o.st.comp_ratio)
    outs = []
    for jit in ("1", "0"):
        env = dict(os.environ, MPC_JIT=jit, MPC_JIT_DEBUG="1", MPC_JIT_CACHE=str(tmp_path / "cache"))
        r = subprocess.run([cli, "-a", "VPC", "-i", p, "-c", cfg_path, "-o", str(tmp_path)], cwd=BIN, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert want in r.stdout, (want, r.stdout[-300:])
        assert ("run-time compilation in the helper process: ok" in r.stderr) == (jit == "1"), r.stderr[-500:]
        outs.append(open(str(tmp_path / "seq_results.csv")).read().strip().split("\n")[-1])      # (the file is appended to, as in the reference)
    assert outs[0] == outs[1] and outs[0].startswith("ds_t,")


@pytest.mark.gpu
def test_cli_line_size_mismatch_is_an_error(cli, configs, traces, tmp_path):
    d = tmp_path / "ds"
    d.mkdir()
    p = traces.save_npy(str(d / "t.npy"), traces.random_u32(10, 32))
    cfg_path = configs.write_config(configs.probe_config(64), str(tmp_path / "c.json"))
    r = run([cli, "-a", "VPC", "-i", p, "-c", cfg_path, "-o", str(tmp_path)])
    assert r.returncode == 1 and "32-byte lines" in r.stdout


@pytest.mark.gpu
def test_per_line_drop_in_matches_batch(cli, oracle, configs, traces, tmp_path):
    """The reference driver's own loop, unchanged (main.cpp:208-248): GetCacheline -> isEnd -> CompressLine per
    line -> GetResult -> Print / PrintDetail, driven (a) by tests/native/perline_probe.cpp through comp::VPC /
    comp::BDI and (b) by `compressor --per-line`.  Every returned size equals the oracle's and the CSV text
    equals what the batch path writes for the same trace (.npy and GPGPU-Sim .log).  The same loop with the
    additive Compressor::SetLineBuffering() (`--line-buffer N`) writes the same rows."""
    host = os.path.join(ROOT, "cal_22-mpc_amd", "host")
    probe = str(tmp_path / "perline_probe")
    srcs = [os.path.join(host, f) for f in sorted(os.listdir(host)) if f.endswith(".cpp") and f != "main.cpp"]
    b = subprocess.run(["hipcc", "-O2", "-std=c++17", "-I", host, "-I", os.path.join(ROOT, "include"), "-o", probe,
                        os.path.join(ROOT, "tests", "native", "perline_probe.cpp"), *srcs,
                        "-L", os.path.join(ROOT, "cal_22-mpc_amd"), "-lmpc_hip",
                        "-Wl,-rpath," + os.path.join(ROOT, "cal_22-mpc_amd")], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-3000:]
    ds = tmp_path / "probe"
    ds.mkdir()
    lines = np.concatenate([traces.structured(1500, 64, seed=31), traces.mixed(600, 64), traces.random_u32(200, 64),
                            traces.zeros(20, 64), traces.word_same(20, 64), traces.bdi_stress(280, 64)])
    lines = lines[np.random.default_rng(12).permutation(len(lines))]
    p = traces.save_npy(str(ds / "trace.npy"), lines)
    cfg = configs.probe_config(64)
    cfg_path = configs.write_config(cfg, str(tmp_path / "probe64.json"))
    for algo, make in (("VPC", lambda: oracle.VpcOracle(cfg)), ("BDI", lambda: oracle.BdiOracle(64))):
        # (a) the C++ classes through the reference-shaped loop
        out = tmp_path / f"pl_{algo}"
        out.mkdir()
        r = run([probe, algo, cfg_path if algo == "VPC" else "-", p, str(out / "r.csv"), str(out / "d.csv"), str(out / "s.bin")])
        assert r.returncode == 0, r.stdout + r.stderr
        assert "lines/s" in r.stdout
        o = make()
        s_ref = o.compress(lines[:-1])[0]                      # LoaderNPY flags the last row as the end
        got = np.fromfile(out / "s.bin", dtype=np.uint16)
        assert len(got) == len(lines) - 1 and (got == s_ref).all()
        # (a') the same loop with Compressor::SetLineBuffering(257): CompressLine returns 0, the rows are the same
        outb = tmp_path / f"plb_{algo}"
        outb.mkdir()
        r = run([probe, algo, cfg_path if algo == "VPC" else "-", p, str(outb / "r.csv"), str(outb / "d.csv"), str(outb / "s.bin"), "257"])
        assert r.returncode == 0, r.stdout + r.stderr
        assert not np.fromfile(outb / "s.bin", dtype=np.uint16).any()
        assert (outb / "r.csv").read_text() == (out / "r.csv").read_text()
        if algo == "VPC":            # (BDIResult has no detail rows)
            assert (outb / "d.csv").read_text() == (out / "d.csv").read_text()
        # (b) the CLI with the per-line loop against the CLI's batch path: identical files
        ob, op = tmp_path / f"batch_{algo}", tmp_path / f"perline_{algo}"
        ob.mkdir()
        op.mkdir()
        extra = ["-c", cfg_path] if algo == "VPC" else []
        rb = run([cli, "-a", algo, "-i", p, "-o", str(ob)] + extra)
        rp = run([cli, "-a", algo, "-i", p, "-o", str(op), "--per-line"] + extra)
        assert rb.returncode == 0 and rp.returncode == 0, rb.stdout + rp.stdout + rp.stderr
        assert rb.stdout == rp.stdout
        opb = tmp_path / f"perline_buffered_{algo}"
        opb.mkdir()
        rpb = run([cli, "-a", algo, "-i", p, "-o", str(opb), "--line-buffer", "1000"] + extra)
        assert rpb.returncode == 0 and rpb.stdout == rb.stdout, rpb.stdout + rpb.stderr
        stem = "probe64" if algo == "VPC" else "BDI"
        assert (ob / f"{stem}_results.csv").read_text() == (op / f"{stem}_results.csv").read_text()
        assert (ob / f"{stem}_results.csv").read_text() == (opb / f"{stem}_results.csv").read_text()
        if algo == "VPC":
            assert (ob / f"{stem}_results_detail.csv").read_text() == (op / f"{stem}_results_detail.csv").read_text()
            # the probe's own rows (workload name "probe_trace") carry the same numbers
            assert (out / "r.csv").read_text().split("\n")[2] == (ob / f"{stem}_results.csv").read_text().split("\n")[2]
            assert (out / "d.csv").read_text().split("\n")[2] == (ob / f"{stem}_results_detail.csv").read_text().split("\n")[2]
    # GPGPU-Sim .log through the per-line loop: the driver-side request filter (main.cpp:222-224)
    types = np.random.default_rng(3).integers(0, 9, 1200)
    plog = traces.write_gpgpusim_log(str(ds / "k.log"), lines[:1200], types)
    ob, op = tmp_path / "batch_log", tmp_path / "perline_log"
    ob.mkdir()
    op.mkdir()
    rb = run([cli, "-a", "VPC", "-i", plog, "-c", cfg_path, "-o", str(ob)])
    rp = run([cli, "-a", "VPC", "-i", plog, "-c", cfg_path, "-o", str(op), "--per-line"])
    assert rb.returncode == 0 and rp.returncode == 0 and rb.stdout == rp.stdout
    assert (ob / "probe64_results_detail.csv").read_text() == (op / "probe64_results_detail.csv").read_text()


@pytest.mark.gpu
def test_cli_apsim_txt(cli, oracle, configs, traces, tmp_path):
    """APSim .txt traffic files through the CLI (reference main.cpp:80-81: 32-byte lines): the data beats of
    handshaking channels, batch and per-line loop alike.  Expected text from the oracle run on
    oracle/apsim_txt.py's reading of the same file (parity unpinned: no fixture exists)."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import apsim_txt as A
    ds = tmp_path / "apsim"
    ds.mkdir()
    beats = np.concatenate([traces.structured(2400, 32, seed=5), traces.counters_u32(800, 32), traces.random_u32(300, 32),
                            traces.zeros(50, 32)])
    beats = beats[np.random.default_rng(4).permutation(len(beats))]
    p = traces.write_apsim_txt(str(ds / "rd.txt"), beats)
    kept = A.lines(p, 32)
    assert len(kept) == len(beats)
    cfg = configs.probe_config(32)
    cfg_path = configs.write_config(cfg, str(tmp_path / "probe32.json"))
    o = oracle.VpcOracle(cfg)
    o.compress(kept)
    row, det = vpc_expected_rows(o, "apsim_rd")
    h1, h2 = vpc_headers(6)
    for extra, sub in (([], "a"), (["--per-line"], "b")):
        out = tmp_path / sub
        out.mkdir()
        r = run([cli, "-a", "VPC", "-i", p, "-c", cfg_path, "-o", str(out)] + extra)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.strip().split("\n")[-1] == "comp.ratio: " + fmt_double(o.st.comp_ratio)
        assert (out / "probe32_results.csv").read_text() == h1 + row + "\n"
        assert (out / "probe32_results_detail.csv").read_text() == h2 + det + "\n"
    pw = traces.write_apsim_txt(str(ds / "wr.txt"), beats[:1500], write_trace=True, final_newline=False)
    kept = A.lines(pw, 32)
    assert 0 < len(kept) < 1500
    r = run([cli, "-a", "BDI", "-i", pw, "-o", str(tmp_path)])
    assert r.returncode == 0, r.stdout + r.stderr
    ob = oracle.BdiOracle(32)
    ob.compress(kept)
    assert r.stdout.strip().split("\n")[-1] == "comp.ratio: " + fmt_double(ob.st.comp_ratio)
