"""GPU parity: the HIP path (through the C ABI of include/mpc_hip.h) against the
CPU oracle on the same inputs -- bit-exact per-line sizes, selected clusters and
the full integer statistics vector.  Runs on the MI355X box (`-m gpu`)."""
import os

import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mpc():
    pkg("build").build_lib()
    return pkg()


def _check_vpc(mpc, oracle, cfg, lines, expect_path=None):
    ev = mpc.VPC(cfg)
    if expect_path is not None:
        assert ev.kernel_path == expect_path
    o = oracle.VpcOracle(cfg)
    s_ref, sel_ref = o.compress(lines)
    s, sel = ev.compress_lines(lines)
    bad = np.nonzero((s != s_ref) | (sel != sel_ref))[0]
    assert bad.size == 0, (f"{bad.size} mismatching lines, first {bad[:5]}: size {s[bad[:5]]} vs {s_ref[bad[:5]]}, "
                           f"sel {sel[bad[:5]]} vs {sel_ref[bad[:5]]}")
    v, v_ref = ev.stats_vector(), o.stats_vector()
    assert v.shape == v_ref.shape
    assert (v == v_ref).all(), f"stats differ at {np.nonzero(v != v_ref)[0][:10]}"
    # derived doubles equal the reference's running doubles (L is a power of two)
    res = ev.result()
    assert res["comp_ratio"] == o.st.comp_ratio
    for c in range(-1, ev.num_modules):
        if ev.line_size & (ev.line_size - 1) == 0:
            assert res["clusters"][c]["mae"] == o.st.mae[c + 1]
            assert res["clusters"][c]["mse"] == o.st.mse[c + 1]
        else:
            # the reference adds a ROUNDED (sum r)/L per line to a running double; the integer sums divide once.
            # The two agree exactly for power-of-two L and to ~1e-15 relative otherwise (deviations).
            assert res["clusters"][c]["mae"] == pytest.approx(o.st.mae[c + 1], rel=1e-12, abs=0)
            assert res["clusters"][c]["mse"] == pytest.approx(o.st.mse[c + 1], rel=1e-12, abs=0)
    ev.close()
    return s, sel


@pytest.mark.parametrize("L", [32, 64, 128])
def test_vpc_fast_path_probe_config(mpc, oracle, configs, traces, L):
    cfg = configs.probe_config(L)
    lines = np.concatenate([
        traces.zeros(70, L), traces.word_same(70, L), traces.random_u32(2000, L),
        traces.sine_f32(2048, L), traces.mixed(2000, L), traces.counters_u32(1000, L),
        traces.structured(6000, L), traces.pointers_u64(500, L) if L >= 8 else traces.zeros(1, L),
        traces.bdi_stress(1400, L),
    ])
    rng = np.random.default_rng(5)
    lines = lines[rng.permutation(len(lines))]
    _check_vpc(mpc, oracle, cfg, lines, expect_path=mpc.MPC_PATH_VPC_FAST)


@pytest.mark.parametrize("L", [32, 64, 128])
def test_vpc_lane_kernel_sequences(mpc, oracle, configs, traces, L):
    """Every module sequence the lane-per-line kernel is unrolled for, with tables
    that need the byte gather (base bytes not simply the previous word) and tables that
    do not, consecutive and plane-0 XOR, with and without AllWordSame."""
    lines = np.concatenate([traces.structured(5000, L, seed=21), traces.mixed(1500, L), traces.random_u32(700, L),
                            traces.sine_f32(1024, L), traces.counters_u32(500, L), traces.zeros(40, L),
                            traces.word_same(40, L)])
    lines = lines[np.random.default_rng(8).permutation(len(lines))]
    prev_word = [max(i - 4, 0) for i in range(L)]
    base1 = [max(i - 1, 0) for i in range(L)]
    base_inword = [4 * (i // 4) if i >= 4 else 0 for i in range(L)]
    diff = [(-3 + (i % 7)) for i in range(L)]
    w3 = [[1.0, 0.5, 1.0, 1.0][i % 4] for i in range(L)]
    w4 = [[1.0, 0.25][i % 2] for i in range(L)]
    az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}
    seqs = [
        [az, aws, configs.one_base(L, 0, True), configs.consecutive_base(L, 0, False),
         configs.diff_base(L, base1, diff, 0, True), configs.weight_base(L, base_inword, w3, 0, False)],
        [az, configs.one_base(L, 0, False), configs.consecutive_base(L, 0, True),
         configs.diff_base(L, prev_word, diff, 0, True), configs.weight_base(L, prev_word, w4, 0, False)],
        [az, aws, configs.one_base(L, 0, True)],
        [az, configs.diff_base(L, base1, diff, 0, False)],
        [az, aws, configs.diff_base(L, prev_word, [1] * L, 0, True)],
        [az, configs.weight_base(L, base1, w3, 0, True)],
        [az, aws, configs.weight_base(L, prev_word, [1.0] * L, 0, True)],
        [az, aws, configs.weight_base(L, prev_word, w4, 0, False)],
    ]
    for mods in seqs:
        _check_vpc(mpc, oracle, configs.make_config(L, mods), lines, expect_path=mpc.MPC_PATH_VPC_FAST)
    # 8-byte elements: predictors that look two words back (BaseIndexTable[i] = i - 8) stay on the
    # fast path; the unrolled 4-predictor sequence, and shorter ones through the run-time loop
    lines64 = np.concatenate([lines[:4000], traces.pointers_u64(2000, L), traces.bdi_stress(1400, L)])
    two_back = [max(i - 8, 0) for i in range(L)]
    d8 = [(3 if i % 8 == 0 else -1 if i % 8 == 1 else 0) for i in range(L)]
    w8 = [[1.0, 0.5, 1.0, 0.5, 1.0, 1.0, 0.5, 1.0][i % 8] for i in range(L)]
    _check_vpc(mpc, oracle, configs.probe_config_u64(L), lines64, expect_path=mpc.MPC_PATH_VPC_FAST)
    for mods in ([az, aws, configs.diff_base(L, two_back, d8, 0, True)],
                 [az, configs.weight_base(L, two_back, w8, 0, False), configs.consecutive_base(L, 0, True)],
                 [az, aws, configs.weight_base(L, two_back, [[2.0, 0.5][i % 2] for i in range(L)], 0, True),
                  configs.diff_base(L, prev_word, diff, 0, False)]):
        _check_vpc(mpc, oracle, configs.make_config(L, mods), lines64, expect_path=mpc.MPC_PATH_VPC_FAST)
    # i - 8 bases whose constants do not repeat every 8 bytes have no built-in form: compiled at creation with a byte gather
    # (the table as constants), the generic kernel without the run-time compiler
    odd = configs.make_config(L, [az, configs.diff_base(L, two_back, diff, 0, True)])
    assert mpc.describe_config(odd)["compiled"] == "at creation"
    _check_vpc(mpc, oracle, odd, lines64[:3000], expect_path=mpc.MPC_PATH_VPC_FAST)
    os.environ["MPC_JIT"] = "0"
    try:
        _check_vpc(mpc, oracle, odd, lines64[:3000], expect_path=mpc.MPC_PATH_VPC_GENERIC)
    finally:
        del os.environ["MPC_JIT"]


def test_vpc_matches_reference_line_vectors(mpc, oracle):
    """Whole lines of the BASELINE workloads against numbers the REFERENCE's compiled stage classes
    produced (tests/golden/ref_line_vectors.json): the HIP path's selected module is the arg-max of the
    reference's leading-zero-row counts with ties to the later module, and its size is that module's
    reference encoder size + id bits, or 8 L + id bits when that is not smaller (VPC.cpp:366-415)."""
    import ref_lines
    cases = ref_lines.load_cases()
    assert len(cases) >= 10
    for name, cfg, lines, c in cases:
        size, sel = ref_lines.expected_sizes(oracle, cfg, lines, c)
        ev = mpc.VPC(cfg)
        assert ev.kernel_path == mpc.MPC_PATH_VPC_FAST, name
        s, k = ev.compress_lines(lines)
        bad = np.nonzero((s != size) | (k != sel))[0]
        assert bad.size == 0, (name, bad[:5], s[bad[:5]], size[bad[:5]], k[bad[:5]], sel[bad[:5]])
        # and in shuffled order / a different position in the wave
        perm = np.random.default_rng(3).permutation(len(lines))
        ev.reset()
        s, k = ev.compress_lines(lines[perm])
        assert (s == size[perm]).all() and (k == sel[perm]).all(), name
        ev.close()


def test_vpc_known_answers_on_gpu(mpc, configs, traces):
    # SURVEY.md 8c known answers, straight from the HIP path
    ev = mpc.VPC(configs.probe_config(64))
    s, sel = ev.compress_lines(traces.zeros(1000))
    assert (s == 3).all() and (sel == 0).all()
    assert repr(ev.result()["comp_ratio"]) == "170.66666666666666"
    ev.reset()
    s, sel = ev.compress_lines(traces.word_same(1000))
    assert (s == 35).all() and (sel == 1).all()
    ev.reset()
    s, sel = ev.compress_lines(traces.random_u32(4096))
    assert (s == 515).all() and (sel == -1).all()
    assert repr(ev.result()["comp_ratio"]) == "0.9941747572815534"
    ev.reset()
    s, sel = ev.compress_lines(traces.sine_f32(4096))
    assert int((s == 515).sum()) == 3456 and int((sel == 5).sum()) == 640
    assert ev.result()["compressed_bits"] == 2105344


def test_vpc_fast_path_variants(mpc, oracle, configs, traces):
    L = 64
    lines = np.concatenate([traces.structured(4000, L, seed=9), traces.mixed(1000, L), traces.random_u32(500, L)])
    # custom id bits, no AllWordSame, non-consecutive XOR on every module
    cfg = configs.make_config(L, [{"name": "AllZero"},
                                  configs.one_base(L, 0, False),
                                  configs.consecutive_base(L, 0, False)],
                              encoding_bits=[1, 2, 5, 7])
    _check_vpc(mpc, oracle, cfg, lines, expect_path=mpc.MPC_PATH_VPC_FAST)
    # windowed tables: previous byte / previous half-word / in-word bases, diffs incl. negatives
    base1 = [max(i - 1, 0) for i in range(L)]
    base2 = [max(i - 2, 0) for i in range(L)]
    base_inword = [4 * (i // 4) if i >= 4 else 0 for i in range(L)]
    diff = [(-3 + (i % 7)) for i in range(L)]
    w3 = [[1.0, 0.5, 1.0, 1.0][i % 4] for i in range(L)]
    w_up = [[2.0, 1.0][i % 2] for i in range(L)]
    w_big = [[256.0, 0.001][i % 2] for i in range(L)]     # shifts >= 8 / <= -8 clear the byte
    w_two = [[2.0, 0.5][i % 2] for i in range(L)]         # two shifted classes (no unshifted one)
    cfg = configs.make_config(L, [{"name": "AllZero"}, {"name": "ByteplaneAllSame"},
                                  configs.diff_base(L, base1, diff, 0, True),
                                  configs.diff_base(L, base2, [0] * L, 0, False),
                                  configs.weight_base(L, base_inword, w3, 0, True),
                                  configs.weight_base(L, base1, w_up, 0, False),
                                  configs.weight_base(L, base2, w_big, 0, True),
                                  configs.weight_base(L, base1, w_two, 0, False)])
    _check_vpc(mpc, oracle, cfg, lines, expect_path=mpc.MPC_PATH_VPC_FAST)
    # a single prediction module, and none at all
    cfg = configs.make_config(L, [{"name": "AllZero"}, {"name": "AllWordSame"}, configs.consecutive_base(L)])
    _check_vpc(mpc, oracle, cfg, lines, expect_path=mpc.MPC_PATH_VPC_FAST)
    cfg = configs.make_config(L, [{"name": "AllZero"}])
    _check_vpc(mpc, oracle, cfg, lines)


@pytest.mark.parametrize("element", [1, 2, 4, 8])
def test_vpc_element_configs(mpc, oracle, configs, traces, element):
    """The authoring helper's configurations for 1/2/4/8-byte elements, 64-byte lines."""
    L = 64
    lines = np.concatenate([traces.structured(3000, L, seed=element), traces.mixed(800, L), traces.pointers_u64(800, L),
                            traces.counters_u32(500, L), traces.random_u32(300, L), traces.zeros(10, L)])
    _check_vpc(mpc, oracle, configs.element_config(L, element), lines, expect_path=mpc.MPC_PATH_VPC_FAST)


@pytest.mark.parametrize("L", [32, 64, 128])
def test_vpc_datatype_model_configs(mpc, oracle, configs, traces, L):
    """The authoring tool's per-datatype models (configs.datatype_config / mpc_config) on the fast kernel,
    against the oracle, on traces of the matching element types."""
    rng = np.random.default_rng(L)
    n = 1500

    def elems(dtype, gen):
        return np.ascontiguousarray(gen.astype(dtype)).view(np.uint8).reshape(n, L)
    per = L // 8
    typed = np.concatenate([
        elems("<i2", (rng.integers(-300, 300, (n, 1)) + np.cumsum(rng.integers(-3, 4, (n, L // 2)), axis=1))),
        elems("<i8", (rng.integers(0, 1 << 40, (n, 1)) + np.cumsum(rng.integers(0, 9, (n, per)), axis=1))),
        elems("<f8", np.cumsum(rng.normal(size=(n, per)), axis=1) * 1e-3 + 1.0),
        elems("<f2", np.cumsum(rng.normal(size=(n, L // 2)), axis=1) * 0.01 + 2.0),
        (rng.integers(0, 2, (n, L)) * rng.integers(0, 2, (n, 1))).astype(np.uint8),
    ])
    lines = np.concatenate([typed, traces.structured(2400, L, seed=5), traces.mixed(600, L), traces.random_u32(200, L),
                            traces.zeros(20, L), traces.word_same(20, L)])
    lines = lines[rng.permutation(len(lines))]
    for dt in configs.DTYPE_BYTES:
        _check_vpc(mpc, oracle, configs.datatype_config(L, dt), lines[:3000], expect_path=mpc.MPC_PATH_VPC_FAST)
    s, sel = _check_vpc(mpc, oracle, configs.mpc_config(L), lines, expect_path=mpc.MPC_PATH_VPC_FAST)
    assert len(set(sel.tolist())) >= 6          # the five models (and the early-outs) are all chosen somewhere


@pytest.mark.parametrize("L", [32, 64, 128])
def test_vpc_fast_path_nonzero_root_and_truncated_scan(mpc, oracle, configs, traces, L):
    """Layouts beyond RootIndex 0 / full plane-major tables that stay on the fast kernel (run-time module loop):
    any RootIndex for OneBase / DiffBase / WeightBase (the residue array is rotated root-first,
    ResidueModule.cpp:24-39) and truncated plane-major scan tables (TableSize < 8 L, the rest of the scanned
    array stays zero, ScanModule.cpp:13-19).  Against the oracle, whose scan stage is pinned by oracle/_ref."""
    rng = np.random.default_rng(100 + L)
    lines = np.concatenate([traces.structured(4000, L, seed=17), traces.mixed(800, L), traces.random_u32(300, L),
                            traces.counters_u32(500, L), traces.zeros(20, L), traces.word_same(20, L)])
    lines = lines[rng.permutation(len(lines))]
    az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}

    def trunc(ts):
        return {"TableSize": ts, "Rows": [i // L for i in range(ts)], "Cols": [i % L for i in range(ts)]}
    prev1 = [max(i - 1, 0) for i in range(L)]
    prev4 = [max(i - 4, 0) for i in range(L)]
    diff = [(-2 + (i % 5)) for i in range(L)]
    w2 = [[1.0, 0.5][i % 2] for i in range(L)]
    for root in (1, 3, 4, 7, L // 2 + 1, L - 1):
        mods = [az, aws, configs.one_base(L, root, True), configs.diff_base(L, prev1, diff, root, False),
                configs.weight_base(L, prev4, w2, root, True), configs.one_base(L, 0, False)]
        _check_vpc(mpc, oracle, configs.make_config(L, mods), lines[:3000], expect_path=mpc.MPC_PATH_VPC_FAST)
    for ts in (8 * L - 24, 6 * L, 4 * L + 7, L, 16, 0):
        mods = [az, aws, configs.one_base(L, 0, True, trunc(ts)), configs.consecutive_base(L, 0, False, trunc(ts)),
                configs.diff_base(L, prev4, diff, 0, True, trunc(ts)), configs.weight_base(L, prev4, w2, 0, True, trunc(ts))]
        _check_vpc(mpc, oracle, configs.make_config(L, mods), lines[:3000], expect_path=mpc.MPC_PATH_VPC_FAST)
    # byte-major scan order (scanned bit i = plane i % 8 of byte i / 8: rows are byte pairs), full and truncated, any root
    def bytemajor(ts=8 * L):
        return {"TableSize": ts, "Rows": [i % 8 for i in range(ts)], "Cols": [i // 8 for i in range(ts)]}
    for ts, root in ((8 * L, 0), (8 * L, 6), (8 * L - 20, 0), (5 * L + 3, 2), (24, 0), (8, 0), (5, 3), (2, 0)):      # (down to a single column)
        mods = [az, aws, configs.one_base(L, root, True, bytemajor(ts)), configs.consecutive_base(L, 0, False, bytemajor(ts)),
                configs.diff_base(L, prev4, diff, root, True, bytemajor(ts)), configs.weight_base(L, prev4, w2, 0, False, bytemajor(ts))]
        _check_vpc(mpc, oracle, configs.make_config(L, mods), lines[:3000], expect_path=mpc.MPC_PATH_VPC_FAST)
    # mixing the two orders has no fast form
    mods = [az, configs.one_base(L, 0, True, bytemajor()), configs.one_base(L, 0, False)]
    _check_vpc(mpc, oracle, configs.make_config(L, mods), lines[:1500], expect_path=mpc.MPC_PATH_VPC_GENERIC)
    # both at once; and tables of different sizes have no fast form
    mods = [az, configs.diff_base(L, prev1, diff, 5, True, trunc(5 * L)), configs.weight_base(L, prev4, w2, 2, False, trunc(5 * L))]
    _check_vpc(mpc, oracle, configs.make_config(L, mods), lines, expect_path=mpc.MPC_PATH_VPC_FAST)
    # tables of different sizes: whole bit planes per module are compiled at creation (a mask per module); any other mix has no fast form
    mods = [az, configs.one_base(L, 0, True, trunc(5 * L)), configs.one_base(L, 0, False, trunc(6 * L))]
    _check_vpc(mpc, oracle, configs.make_config(L, mods), lines[:1500], expect_path=mpc.MPC_PATH_VPC_FAST)
    mods = [az, configs.one_base(L, 0, True, trunc(5 * L)), configs.one_base(L, 0, False, trunc(6 * L + 3))]
    _check_vpc(mpc, oracle, configs.make_config(L, mods), lines[:1500], expect_path=mpc.MPC_PATH_VPC_GENERIC)
    os.environ["MPC_JIT"] = "0"                  # without the run-time compiler the different whole-plane sizes take the generic kernel
    try:
        mods = [az, configs.one_base(L, 0, True, trunc(5 * L)), configs.one_base(L, 0, False, trunc(6 * L))]
        assert mpc.describe_config(configs.make_config(L, mods))["path"] == "generic"
        _check_vpc(mpc, oracle, configs.make_config(L, mods), lines[:1500], expect_path=mpc.MPC_PATH_VPC_GENERIC)
    finally:
        del os.environ["MPC_JIT"]


@pytest.mark.parametrize("L", [32, 64, 128])
def test_vpc_general_layout_twins(mpc, oracle, configs, traces, L):
    """RootIndex 1..15 and truncated plane-major scan tables (16 <= TableSize < 8 L: whole bit planes through one mask,
    tables cut inside a plane through per-word masks) keep the UNROLLED
    kernels (their general-layout twins, vpc_lane_gen_kernel: the same row-0 prefilters and line ring; round 2 sent these
    to the run-time module loop at 0.27-0.33 of the peak).  Module sequences with an instantiation: the probe's four
    predictors with 4- and 8-byte-back tables, their non-periodic form (a root outside word 0 breaks the period), single
    models.  With per-line outputs and in statistics-only mode; a root above 15 or a table without a complete first row still takes
    the run-time loop.  ResidueModule.cpp:24-39 (root first), ScanModule.cpp:13-19 (untouched cells stay zero)."""
    rng = np.random.default_rng(300 + L)
    lines = np.concatenate([traces.structured(5000, L, seed=23), traces.mixed(2500, L), traces.random_u32(800, L),
                            traces.sine_f32(1024, L), traces.counters_u32(500, L), traces.zeros(70, L), traces.word_same(70, L)])
    lines = lines[rng.permutation(len(lines))]
    az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}

    def trunc(ts):
        return None if ts is None else {"TableSize": ts, "Rows": [i // L for i in range(ts)], "Cols": [i % L for i in range(ts)]}
    prev4 = [max(i - 4, 0) for i in range(L)]
    prev8 = [max(i - 8, 0) for i in range(L)]
    w2 = [[1.0, 0.5][i % 2] for i in range(L)]
    d1 = [1 if i % 4 == 0 else 0 for i in range(L)]

    def probe(roots, ts, prev=prev4):
        s = trunc(ts)
        return configs.make_config(L, [az, aws, configs.one_base(L, roots[0], True, s), configs.consecutive_base(L, 0, True, s),
                                       configs.diff_base(L, prev, d1, roots[1], False, s), configs.weight_base(L, prev, w2, roots[2], True, s)])

    def check(cfg, general, n=len(lines)):
        d = mpc.describe_config(cfg)
        assert d["path"] == "fast" and d["general_layout"] == ("yes" if general else "no"), d
        assert d["sequence"] == ("unrolled" if general else "run-time loop"), d
        _check_vpc(mpc, oracle, cfg, lines[:n], expect_path=mpc.MPC_PATH_VPC_FAST)
        ev = mpc.VPC(cfg)                                   # the kernels without per-line outputs
        ev.compress_lines(lines[:n], want_sizes=False, want_selected=False)
        o = oracle.VpcOracle(cfg)
        o.compress(lines[:n])
        assert (ev.stats_vector() == o.stats_vector()).all()
        ev.close()

    # roots inside word 0 keep the periodic tables; every whole-plane table size; both at once
    for roots, ts in (((5, 3, 2), None), ((0, 0, 0), 6 * L), ((15, 1, 3), 7 * L), ((1, 0, 0), L), ((12, 2, 1), 3 * L),
                      ((4, 3, 0), None), ((8, 0, 2), 2 * L), ((0, 3, 3), 4 * L), ((2, 0, 0), 5 * L)):
        check(probe(roots, ts), True, 6000)
    check(probe((7, 5, 6), 5 * L, prev8), True, 6000)        # 8-byte-back tables, roots inside the first element
    check(probe((3, 6, 9), None), True, 6000)                # roots outside word 0: the non-periodic instantiation
    check(probe((9, 13, 15), 6 * L), True)
    # single models
    for mods in ([az, aws, configs.one_base(L, 11, True, trunc(6 * L))], [az, configs.diff_base(L, prev4, d1, 2, True)],
                 [az, aws, configs.weight_base(L, prev8, w2, 5, False, trunc(3 * L))], [az, aws, configs.one_base(L, 7, False), configs.consecutive_base(L, 0, True)]):
        check(configs.make_config(L, mods), True, 5000)
    # tables cut inside a bit plane (per-word masks), down to a single complete row
    for roots, ts in (((0, 0, 0), 6 * L + 8), ((6, 2, 1), 4 * L + 7), ((0, 0, 0), 8 * L - 24), ((0, 3, 0), 16), ((9, 0, 0), L + 20)):
        check(probe(roots, ts), True, 5000)
    # a root above 15: no built-in twin takes it; the general-layout kernel is compiled at creation with the roots as constants
    # (the rotation then reaches beyond row 0, and the prefilter drops natural byte 15 for the raw root)
    far = (((16, 0, 0), None), ((L - 1, 17, L // 2 + 1), None), ((5, 20, 0), 6 * L), ((19, L - 2, 16), 4 * L + 7))
    for roots, ts in (far if L == 64 else far[1:3]):           # (each case is a compilation: all of them at 64-byte lines only)
        cfg = probe(roots, ts)
        assert mpc.describe_config(cfg)["compiled"] == "at creation"
        check(cfg, True, 5000)
        ev = mpc.VPC(cfg)
        assert ev.kernel_form.startswith("unrolled, compiled at creation"), ev.kernel_form
        ev.close()
    # every module its own number of bit planes (no built-in kernel, not the run-time loop either: compiled at creation with a
    # mask per module; the winner's mask goes with it to the XOR stage) -- round 2 ran these on the generic kernel
    planes = (((8, 6, 4, 7), (0, 3, 0)), ((3, 8, 8, 1), (5, 0, 2)), ((8, 8, 8, 2), (0, 0, 0)))
    for sizes, roots in (planes if L == 64 else planes[:2]):
        mods = [az, aws, configs.one_base(L, roots[0], True, trunc(sizes[0] * L)), configs.consecutive_base(L, 0, True, trunc(sizes[1] * L)),
                configs.diff_base(L, prev4, d1, roots[1], False, trunc(sizes[2] * L)), configs.weight_base(L, prev4, w2, roots[2], True, trunc(sizes[3] * L))]
        cfg = configs.make_config(L, mods)
        d = mpc.describe_config(cfg)
        assert d["path"] == "fast" and d["compiled"] == "at creation", d
        check(cfg, True, 6000)
    # outside the unrolled kernels' reach: the run-time loop
    check(probe((0, 0, 0), 8), False, 3000)


@pytest.mark.parametrize("L", [32, 64, 128])
def test_vpc_sequences_compiled_at_creation(mpc, oracle, configs, traces, L, tmp_path, monkeypatch):
    """A module sequence without a built-in unrolled instantiation gets one when the handle is created: hiprtc compiles the
    lane kernel's own source with the sequence as template arguments (csrc/mpc_jit.h; round 2 ran such sequences on the
    run-time loop at 0.3 of the peak).  Results against the oracle with per-line outputs and in statistics-only mode; the
    handle says which form it launches; the code object is cached (second handle: from the cache); MPC_JIT=0 keeps the
    run-time loop, with the same results."""
    monkeypatch.setenv("MPC_JIT_CACHE", str(tmp_path / "jit"))
    monkeypatch.delenv("MPC_JIT", raising=False)
    rng = np.random.default_rng(500 + L)
    lines = np.concatenate([traces.structured(4000, L, seed=29), traces.mixed(2500, L), traces.random_u32(600, L),
                            traces.sine_f32(1024, L), traces.counters_u32(400, L), traces.zeros(70, L), traces.word_same(70, L)])
    lines = lines[rng.permutation(len(lines))]
    az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}

    def trunc(ts):
        return None if ts is None else {"TableSize": ts, "Rows": [i // L for i in range(ts)], "Cols": [i % L for i in range(ts)]}
    prev1 = [max(i - 1, 0) for i in range(L)]
    prev4 = [max(i - 4, 0) for i in range(L)]
    prev8 = [max(i - 8, 0) for i in range(L)]
    diff = [(-2 + (i % 5)) for i in range(L)]
    w2 = [[1.0, 0.5][i % 2] for i in range(L)]
    w3 = [[2.0, 0.25][i % 2] for i in range(L)]              # two shifted classes
    seqs = {
        "OB DF WT OB": lambda s, r: [az, aws, configs.one_base(L, r[0], True, s), configs.diff_base(L, prev1, diff, r[1], False, s),
                                     configs.weight_base(L, prev4, w2, r[2], True, s), configs.one_base(L, 0, False, s)],
        "CS WT2 OB": lambda s, r: [az, configs.consecutive_base(L, 0, True, s), configs.weight_base(L, prev4, w3, r[1], False, s),
                                   configs.one_base(L, r[0], True, s)],
        "8 modules": lambda s, r: [az, aws, configs.diff_base(L, prev8, [1] * L, r[1], True, s), configs.one_base(L, r[0], False, s),
                                   configs.consecutive_base(L, 0, False, s), configs.weight_base(L, prev8, w2, r[2], True, s),
                                   configs.diff_base(L, prev4, diff, 0, False, s), configs.weight_base(L, prev1, w2, 0, False, s),
                                   configs.one_base(L, 0, True, s), configs.consecutive_base(L, 0, True, s)],
    }

    def run(cfg, form_prefix):
        ev = mpc.VPC(cfg)
        assert ev.kernel_path == mpc.MPC_PATH_VPC_FAST and ev.kernel_form.startswith(form_prefix), ev.kernel_form
        o = oracle.VpcOracle(cfg)
        s_ref, k_ref = o.compress(lines)
        s, k = ev.compress_lines(lines)
        assert (s == s_ref).all() and (k == k_ref).all() and (ev.stats_vector() == o.stats_vector()).all()
        ev.reset()
        ev.compress_lines(lines, want_sizes=False, want_selected=False)
        assert (ev.stats_vector() == o.stats_vector()).all()
        form = ev.kernel_form
        ev.close()
        return form

    for name, make in seqs.items():
        for ts, roots in ((None, (0, 0, 0)), (6 * L, (5, 3, 2))):          # plain layout, general layout
            cfg = configs.make_config(L, make(trunc(ts), roots))
            # (128-byte lines, 8 modules: 11 clusters x 1030 bins leave no room for the rings of 8 waves -- the kernel is compiled
            # for a smaller workgroup, mpc_vpc_lane_ring_plan)
            d = mpc.describe_config(cfg)
            assert d["sequence"] == "unrolled" and d["compiled"] == "at creation" and d["general_layout"] == ("yes" if ts else "no"), (name, d)
            run(cfg, "unrolled, compiled at creation")       # (compiled now, unless an earlier test of this process had the same shape)
            if L == 64 or name == "OB DF WT OB":
                assert run(cfg, "unrolled, compiled at creation") == "unrolled, compiled at creation (from the cache)"
    # the byte-major scan order (rows = byte pairs): no built-in kernel has it; with every RootIndex 0 the unrolled kernels are
    # compiled for it (its own selector, row-0 prefilter, certificate and encoder), full and truncated tables; a root elsewhere
    # keeps the run-time loop
    def bytemajor(ts=8 * L):
        return {"TableSize": ts, "Rows": [i % 8 for i in range(ts)], "Cols": [i // 8 for i in range(ts)]}
    d1 = [1 if i % 4 == 0 else 0 for i in range(L)]
    for ts in ((8 * L, 8 * L - 20, 5 * L + 3, 24, 16) if L == 64 else (8 * L, 5 * L + 3)):      # (each case is a compilation)
        sb = bytemajor(ts)
        cfg = configs.make_config(L, [az, aws, configs.one_base(L, 0, True, sb), configs.consecutive_base(L, 0, True, sb),
                                      configs.diff_base(L, prev4, d1, 0, False, sb), configs.weight_base(L, prev4, w2, 0, True, sb)])
        d = mpc.describe_config(cfg)
        assert d["scan_order"] == "byte-major" and d["sequence"] == "unrolled" and d["compiled"] == "at creation", d
        run(cfg, "unrolled, compiled at creation")
    sb = bytemajor()
    run(configs.make_config(L, [az, configs.one_base(L, 0, True, sb), configs.diff_base(L, prev1, diff, 0, False, sb),
                                configs.weight_base(L, prev4, w3, 0, False, sb), configs.one_base(L, 0, False, sb)]), "unrolled, compiled at creation")
    run(configs.make_config(L, [az, aws, configs.consecutive_base(L, 0, True, sb)]), "unrolled, compiled at creation")
    run(configs.make_config(L, [az, aws, configs.one_base(L, 6, True, sb), configs.consecutive_base(L, 0, False, sb)]), "run-time loop")
    run(configs.make_config(L, [az, aws, configs.one_base(L, 0, True, bytemajor(8)), configs.consecutive_base(L, 0, False, bytemajor(8))]), "run-time loop")
    # base tables that are not windowed (a base byte anywhere in the line: 16 / 12 bytes back, byte 0 for all, reversed):
    # gathered with the table as compile-time constants -- round 2 ran these on the generic kernel
    prev16 = [max(i - 16, 0) for i in range(L)]
    prev12 = [max(i - 12, 0) for i in range(L)]
    rev = [L - 1 - i for i in range(L)]
    for mods in ([az, aws, configs.one_base(L, 0, True), configs.consecutive_base(L, 0, True), configs.diff_base(L, prev16, diff, 0, False),
                  configs.weight_base(L, prev12, w2, 0, True)],
                 [az, configs.diff_base(L, [0] * L, [1] * L, 0, True), configs.weight_base(L, prev4, w2, 3, False)],
                 [az, aws, configs.weight_base(L, rev, w3, 5, True, trunc(6 * L)), configs.one_base(L, 3, False, trunc(6 * L)),
                  configs.diff_base(L, prev16, diff, 9, True, trunc(6 * L))]):
        cfg = configs.make_config(L, mods)
        d = mpc.describe_config(cfg)
        assert d["path"] == "fast" and d["compiled"] == "at creation", d
        run(cfg, "unrolled, compiled at creation")
    # weight tables with more than two shift distances (2^-9 .. 2^9 per byte position; shifts of 8 and more predict 0): the
    # predicted word is assembled from the table as compile-time constants -- alone, with a gathered base, roots and truncation
    w4 = [[1.0, 0.5, 0.25, 2.0][i % 4] for i in range(L)]
    w19 = [float(2.0 ** ((i * 7) % 19 - 9)) for i in range(L)]
    for mods in ([az, aws, configs.weight_base(L, prev4, w4, 0, True), configs.one_base(L, 0, False)],
                 [az, configs.weight_base(L, prev16, w19, 6, False), configs.consecutive_base(L, 0, True), configs.weight_base(L, prev1, w4, 0, True)],
                 [az, aws, configs.diff_base(L, prev4, diff, 2, True, trunc(5 * L)), configs.weight_base(L, prev4, w19, 0, False, trunc(5 * L))]):
        cfg = configs.make_config(L, mods)
        d = mpc.describe_config(cfg)
        assert d["path"] == "fast" and d["compiled"] == "at creation", d
        run(cfg, "unrolled, compiled at creation")
    if L == 64:      # the longest sequence that is compiled (12 prediction modules; the kernel spills registers but beats the loop)
        twelve = seqs["8 modules"](None, (0, 0, 0)) + [configs.one_base(L, 0, False), configs.diff_base(L, prev1, diff, 0, True),
                                                        configs.weight_base(L, prev4, w2, 0, False), configs.consecutive_base(L, 0, True)]
        run(configs.make_config(L, twelve), "unrolled, compiled at creation")
        run(configs.make_config(L, twelve + [configs.one_base(L, 0, True)]), "run-time loop")
    # the paper figure's five models: a built-in sequence, but at 128-byte lines its 8 clusters x 1029 bins leave 1 KiB too
    # little for the rings of the built-in 8-wave workgroup -- compiled at creation for 7 waves instead of the run-time loop
    cfg5 = configs.mpc_config(L)
    assert mpc.describe_config(cfg5)["compiled"] == ("at creation" if L == 128 else "built in")
    run(cfg5, "unrolled, compiled at creation" if L == 128 else "unrolled")
    # switched off: the run-time module loop, same results
    monkeypatch.setenv("MPC_JIT", "0")
    cfg = configs.make_config(L, seqs["OB DF WT OB"](None, (0, 0, 0)))
    assert mpc.describe_config(cfg)["sequence"] == "run-time loop"
    assert run(cfg, "run-time loop") == "run-time loop"


def test_vpc_many_modules_large_histogram(mpc, oracle, configs, traces):
    """16 modules at 128-byte lines: 17 clusters x 1030 bins = 70 KB of LDS histogram, more than a
    kernel's default 64 KiB (fast kernel with the run-time module loop, and the generic kernel)."""
    L = 128
    lines = np.concatenate([traces.structured(2500, L, seed=3), traces.mixed(600, L), traces.random_u32(300, L),
                            traces.zeros(10, L)])
    mods = [{"name": "AllZero"}, {"name": "AllWordSame"}]
    for i in range(14):
        mods.append(configs.one_base(L, 0, bool(i & 1)) if i % 3 == 0 else
                    configs.consecutive_base(L, 0, bool(i & 1)) if i % 3 == 1 else
                    configs.diff_base(L, [max(j - 1 - (i % 4), 0) if j % 4 > i % 4 else max(4 * (j // 4) - 1, 0) for j in range(L)],
                                      [i] * L, 0, bool(i & 1)))
    cfg = configs.make_config(L, mods)
    _check_vpc(mpc, oracle, cfg, lines, expect_path=mpc.MPC_PATH_VPC_FAST)
    perm = [int(x) for x in np.random.default_rng(1).permutation(8 * L)]   # a permuted scan table has no fast form
    mods[2] = configs.one_base(L, 0, True, {"TableSize": 8 * L, "Rows": [p // L for p in perm], "Cols": [p % L for p in perm]})
    _check_vpc(mpc, oracle, configs.make_config(L, mods), lines[:1500], expect_path=mpc.MPC_PATH_VPC_GENERIC)


def test_vpc_more_than_sixteen_modules(mpc, oracle, configs, traces):
    """24 modules (22 prediction modules) at 64-byte lines: above the former limit of 16; both kernels."""
    L = 64
    lines = np.concatenate([traces.structured(2400, L, seed=13), traces.mixed(500, L), traces.random_u32(200, L)])
    mods = [{"name": "AllZero"}, {"name": "AllWordSame"}]
    for i in range(22):
        base = [max(j - 1 - (i % 3), 0) if j % 4 > i % 3 else max(4 * (j // 4) - 1, 0) for j in range(L)]
        mods.append([configs.one_base(L, 0, bool(i & 1)), configs.consecutive_base(L, 0, bool(i & 2)),
                     configs.diff_base(L, base, [i - 9] * L, 0, bool(i & 1)),
                     configs.weight_base(L, base, [[1.0, 0.5, 2.0][(i + j) % 2] for j in range(L)], 0, bool(i & 4))][i % 4])
    enc = [int(x) for x in np.random.default_rng(2).integers(0, 12, len(mods) + 1)]
    _check_vpc(mpc, oracle, configs.make_config(L, mods, enc), lines, expect_path=mpc.MPC_PATH_VPC_FAST)
    perm = [int(x) for x in np.random.default_rng(1).permutation(8 * L)]
    mods[5] = configs.one_base(L, 9, False, {"TableSize": 8 * L, "Rows": [p // L for p in perm], "Cols": [p % L for p in perm]})
    _check_vpc(mpc, oracle, configs.make_config(L, mods, enc), lines[:1200], expect_path=mpc.MPC_PATH_VPC_GENERIC)


def _random_windowed_config(configs, rng, L):
    """Random fast-path configuration: 1..6 prediction modules of random kinds; windowed base
    tables (base byte in the own or the previous 32-bit word, never ahead of the byte), random
    diffs, weights with at most two shift distances, random XOR flavour, optional AllWordSame,
    random id bits."""
    def windowed_base():
        b = [0] * L
        for i in range(1, L):
            lo = max(4 * (i // 4) - 4, 0)
            b[i] = int(rng.integers(lo, 4 * (i // 4) + 4))
        return b
    mods = [{"name": "AllZero"}]
    if rng.integers(0, 2):
        mods.append({"name": "AllWordSame"})
    for _ in range(int(rng.integers(1, 7))):
        kind = int(rng.integers(0, 4))
        cx = bool(rng.integers(0, 2))
        if kind == 0:
            mods.append(configs.one_base(L, 0, cx))
        elif kind == 1:
            mods.append(configs.consecutive_base(L, 0, cx))
        elif kind == 2:
            diff = [int(x) for x in rng.integers(-128, 128, L)] if rng.integers(0, 2) else [int(rng.integers(-3, 4))] * L
            mods.append(configs.diff_base(L, windowed_base(), diff, 0, cx))
        else:
            pool = [[1.0, 0.5], [1.0, 0.25], [2.0, 1.0], [2.0, 0.5], [1.0, 1.0], [0.125, 4.0], [1.0, 300.0]][int(rng.integers(0, 7))]
            w = [pool[int(rng.integers(0, 2))] for _ in range(L)]
            mods.append(configs.weight_base(L, windowed_base(), w, 0, cx))
    enc = [int(x) for x in rng.integers(0, 9, len(mods) + 1)] if rng.integers(0, 2) else None
    return configs.make_config(L, mods, enc)


@pytest.mark.parametrize("L", [32, 64, 128])
def test_vpc_fast_path_random_configs(mpc, oracle, configs, traces, L):
    """Randomly drawn fast-path configurations (mostly run-time module sequences) on a mixed
    trace, bit-exact against the oracle."""
    rng = np.random.default_rng(1000 + L)
    lines = np.concatenate([traces.structured(2500, L, seed=L), traces.mixed(800, L), traces.random_u32(300, L),
                            traces.sine_f32(512, L), traces.zeros(20, L), traces.word_same(20, L)])
    lines = lines[rng.permutation(len(lines))]
    for _ in range(10):
        cfg = _random_windowed_config(configs, rng, L)
        _check_vpc(mpc, oracle, cfg, lines, expect_path=mpc.MPC_PATH_VPC_FAST)


@pytest.mark.parametrize("L", [32, 64, 128, 48])
def test_vpc_generic_path(mpc, oracle, configs, traces, L):
    rng = np.random.default_rng(L)
    n = 8 * L
    perm = rng.permutation(n)
    scan = {"TableSize": n - 16, "Rows": [int(p) // L for p in perm[: n - 16]], "Cols": [int(p) % L for p in perm[: n - 16]]}
    rb = [int(x) for x in rng.integers(0, L, L)]
    rd = [int(x) for x in rng.integers(-300, 300, L)]
    rw = [float(2.0 ** int(x)) for x in rng.integers(-9, 10, L)]
    mods = [{"name": "AllZero"}, {"name": "AllWordSame"},
            configs.one_base(L, root=5, consecutive_xor=True),
            configs.consecutive_base(L, 0, False, scan=scan),
            configs.diff_base(L, rb, rd, root=3, consecutive_xor=True),
            configs.weight_base(L, rb, rw, root=L - 1, consecutive_xor=False, scan=scan)]
    cfg = configs.make_config(L, mods)
    if L == 48:
        lines = rng.integers(0, 256, (1500, L), dtype=np.uint8)
        lines[::3] //= 64
        lines[::5] = 0
    else:
        lines = np.concatenate([traces.structured(2400, L, seed=3), traces.random_u32(300, L),
                                traces.zeros(20, L), traces.word_same(20, L), traces.mixed(400, L)])
    _check_vpc(mpc, oracle, cfg, lines, expect_path=mpc.MPC_PATH_VPC_GENERIC)


def test_generic_and_fast_agree(mpc, configs, traces):
    # the same semantics through both kernels: OneBase(root 0) spelled as a
    # DiffBase with base 0 / diff 0 everywhere is not "windowed", so that
    # configuration takes the generic kernel and must give identical results
    L = 64
    lines = np.concatenate([traces.structured(3000, L, seed=11), traces.sine_f32(1000, L)])
    cfg = configs.probe_config(L)
    fast = mpc.VPC(cfg)
    assert fast.kernel_path == mpc.MPC_PATH_VPC_FAST
    s1, c1 = fast.compress_lines(lines)
    eq = configs.make_config(L, [{"name": "AllZero"}, {"name": "AllWordSame"},
                                 configs.diff_base(L, [0] * L, [0] * L, 0, True),
                                 cfg["modules"]["3"], cfg["modules"]["4"], cfg["modules"]["5"]])
    jit = mpc.VPC(eq)                         # ... compiled at creation with the base table as constants (a byte gather)
    assert jit.kernel_path == mpc.MPC_PATH_VPC_FAST and jit.kernel_form.startswith("unrolled, compiled at creation"), jit.kernel_form
    s3, c3 = jit.compress_lines(lines)
    assert (s1 == s3).all() and (c1 == c3).all()
    os.environ["MPC_JIT"] = "0"                 # ... and without the run-time compiler on the generic kernel
    try:
        gen = mpc.VPC(eq)
    finally:
        del os.environ["MPC_JIT"]
    assert gen.kernel_path == mpc.MPC_PATH_VPC_GENERIC and "windowed" in gen.path_reason and fast.path_reason == ""
    s2, c2 = gen.compress_lines(lines)
    assert (s1 == s2).all() and (c1 == c2).all()
    assert (fast.stats_vector() == gen.stats_vector()).all()


@pytest.mark.parametrize("L", [32, 64, 128])
def test_bdi(mpc, oracle, traces, L):
    lines = np.concatenate([traces.bdi_stress(5600, L), traces.random_u32(500, L), traces.zeros(30, L),
                            traces.structured(2400, L), traces.pointers_u64(1000, L), traces.mixed(1000, L),
                            traces.bdi_screen_stress(14400, L), traces.bdi_screen_stress(7200, L, seed=9)[::-1]])
    ev = mpc.BDI(L)
    o = oracle.BdiOracle(L)
    s_ref, sel_ref = o.compress(lines)
    s, sel = ev.compress_lines(lines)
    bad = np.nonzero((s != s_ref) | (sel != sel_ref))[0]
    assert bad.size == 0, f"{bad.size} mismatches, first {bad[:5]}: {s[bad[:5]]} vs {s_ref[bad[:5]]}"
    assert (ev.stats_vector() == o.stats_vector()).all()
    assert len(np.unique(sel_ref)) == 9
    if L == 128:
        ev.reset()
        s, sel = ev.compress_lines(traces.pointers_u64(4096, 128))
        assert (s == 564).all() and (sel == 4).all()


@pytest.mark.parametrize("L", [8, 24, 48, 96, 120])
def test_baselines_any_line_size(mpc, oracle, traces, L):
    """BDI / FPC / BPC at line sizes other than 32 / 64 / 128 (the reference takes whatever line size the
    loader reports): the loop kernel, against the oracle."""
    rng = np.random.default_rng(L)
    lines = np.concatenate([traces.bdi_stress(2800, 128)[:, :L], traces.structured(2400, 128)[:, :L],
                            traces.random_u32(300, 128)[:, :L], np.zeros((20, L), np.uint8),
                            traces.counters_u32(500, 128)[:, :L]])
    lines = np.ascontiguousarray(lines[rng.permutation(len(lines))])
    ev, o = mpc.BDI(L), oracle.BdiOracle(L)
    s_ref, sel_ref = o.compress(lines)
    s, sel = ev.compress_lines(lines)
    assert (s == s_ref).all() and (sel == sel_ref).all() and (ev.stats_vector() == o.stats_vector()).all()
    ev.close()
    ev, o = mpc.FPC(L), oracle.FpcOracle(L)
    s, _ = ev.compress_lines(lines)
    assert (s == o.compress(lines)).all() and (ev.stats_vector() == o.stats_vector()).all()
    ev.close()
    ev, o = mpc.BPC(L), oracle.BpcOracle(L)
    s, _ = ev.compress_lines(lines)
    assert (s == o.compress(lines)).all() and (ev.stats_vector() == o.stats_vector()).all()
    ev.close()


@pytest.mark.parametrize("L", [32, 64, 128])
def test_fpc(mpc, oracle, traces, L):
    """FPC baseline against the oracle's source-reading restatement (parity unpinned: the
    reference has no fixture for it): every prefix, zero runs of every length incl. to the end of
    the line, sign-extension boundaries."""
    rng = np.random.default_rng(L)
    n = 6000
    special = np.array([0, 1, 7, 8, 0xFFFFFFF8, 0xFFFFFFF7, 0x7F, 0x80, 0xFFFFFF80, 0xFFFFFF7F, 0x7FFF, 0x8000,
                        0xFFFF8000, 0xFFFF7FFF, 0x10000, 0x12340000, 0x007F007F, 0xFF80FF80, 0x0080007F, 0xFF7FFF80,
                        0x007FFF80, 0xFF80007F, 0xABABABAB, 0x00000100, 0x01010101, 0x12345678, 0xFFFFFFFF, 0x80000000],
                       dtype=np.uint32)
    words = special[rng.integers(0, len(special), (n, L // 4))]
    words[rng.random((n, L // 4)) < 0.35] = 0                      # zero runs
    words[::7, -3:] = 0                                            # runs that reach the end of the line
    words[1::7, :] = 0
    lines = np.concatenate([words.astype("<u4").view(np.uint8).reshape(n, L), traces.structured(3000, L),
                            traces.mixed(1000, L), traces.random_u32(500, L), traces.zeros(50, L)])
    ev, o = mpc.FPC(L), oracle.FpcOracle(L)
    assert ev.kernel_path == mpc.MPC_PATH_FPC
    s_ref = o.compress(lines)
    s, sel = ev.compress_lines(lines)
    bad = np.nonzero(s != s_ref)[0]
    assert bad.size == 0, f"{bad.size} mismatches, first {bad[:5]}: {s[bad[:5]]} vs {s_ref[bad[:5]]}"
    assert (sel == 0).all()
    assert (ev.stats_vector() == o.stats_vector()).all()
    assert (o.stats_vector()[3:] > 0).all()                        # every prefix occurs
    assert ev.result()["comp_ratio"] == o.st.comp_ratio and ev.result()["total_words"] == o.st.total_words


@pytest.mark.parametrize("L", [32, 64, 128])
def test_bpc(mpc, oracle, traces, L):
    """BPC baseline against the oracle's source-reading restatement (parity unpinned): ramps and
    constant lines (zero runs, all-ones planes), sparse deltas (single / adjacent ones), wrap-around
    deltas (borrow plane), random and structured data."""
    rng = np.random.default_rng(L + 1)
    n, W = 6000, L // 4
    base = rng.integers(0, 1 << 32, (n, 1), dtype=np.uint64)
    step = rng.choice(np.array([0, 1, 2, 3, 255, 256, 0xFFFFFFFF, 0xFFFFFF00, 65536, 0x80000000], dtype=np.uint64), (n, 1))
    words = (base + step * np.arange(W, dtype=np.uint64)[None, :]) & np.uint64(0xFFFFFFFF)
    bump = rng.random((n, W)) < 0.08                              # sparse extra deltas
    words = (words + bump * rng.integers(1, 4, (n, W)).astype(np.uint64)) & np.uint64(0xFFFFFFFF)
    words[::9] = np.uint64(0)
    words[1::9, :] = words[1::9, :1]
    lines = np.concatenate([words.astype("<u4").view(np.uint8).reshape(n, L), traces.structured(3000, L),
                            traces.mixed(1000, L), traces.random_u32(500, L), traces.counters_u32(500, L),
                            traces.pointers_u64(500, L)])
    ev, o = mpc.BPC(L), oracle.BpcOracle(L)
    assert ev.kernel_path == mpc.MPC_PATH_BPC
    s_ref = o.compress(lines)
    s, sel = ev.compress_lines(lines)
    bad = np.nonzero(s != s_ref)[0]
    assert bad.size == 0, f"{bad.size} mismatches, first {bad[:5]}: {s[bad[:5]]} vs {s_ref[bad[:5]]}"
    assert (ev.stats_vector() == o.stats_vector()).all()
    counts = o.stats_vector()[4:]
    assert counts[5] == 0 and (np.delete(counts, [5] + ([] if L == 128 else [6])) > 0).all()   # ZeroDBP is never used
    assert ev.result()["comp_ratio"] == o.st.comp_ratio and ev.result()["total_words"] == 33 * len(lines)


def test_edge_cases(mpc, oracle, configs, traces):
    cfg = configs.probe_config(64)
    ev = mpc.VPC(cfg)
    # empty batch
    s, sel = ev.compress_lines(np.zeros((0, 64), dtype=np.uint8))
    assert len(s) == 0 and ev.stats_vector()[0] == 0
    # ragged tails: sizes that do not fill a wave / a workgroup
    o = oracle.VpcOracle(cfg)
    for n in (1, 3, 15, 17, 63, 65, 255, 257, 1000):
        ev.reset(); o.reset()
        lines = traces.structured(n, 64, seed=n)
        s, sel = ev.compress_lines(lines)
        sr, cr = o.compress(lines)
        assert (s == sr).all() and (sel == cr).all()
        assert (ev.stats_vector() == o.stats_vector()).all()
    # statistics accumulate across calls, merge and set behave like integer sums
    ev.reset(); o.reset()
    a, b = traces.mixed(777, 64), traces.structured(555, 64)
    ev.compress_lines(a, want_sizes=False, want_selected=False)
    ev.compress_lines(b, want_sizes=True, want_selected=False)
    o.compress(a); o.compress(b)
    v = ev.stats_vector()
    assert (v == o.stats_vector()).all()
    ev.stats_merge(v)
    assert (ev.stats_vector() == 2 * v).all()
    ev.stats_set(v)
    assert (ev.stats_vector() == v).all()
    # wrong line size is an error code, not a crash
    with pytest.raises(ValueError):
        ev.compress_lines(np.zeros((4, 32), dtype=np.uint8))
    # the baselines on the same ragged sizes, statistics accumulated over the calls
    for make, ref, two in ((mpc.BDI, oracle.BdiOracle, True), (mpc.FPC, oracle.FpcOracle, False),
                           (mpc.BPC, oracle.BpcOracle, False)):
        for L in (32, 128):
            be, bo = make(L), ref(L)
            assert be.compress_lines(np.zeros((0, L), dtype=np.uint8))[0].size == 0
            for n in (1, 2, 63, 64, 65, 257, 1000):
                lines = traces.structured(n, L, seed=n + L)
                got = be.compress_lines(lines)[0]
                want = bo.compress(lines)
                assert (got == (want[0] if two else want)).all(), (make.__name__, L, n)
            assert (be.stats_vector() == bo.stats_vector()).all(), (make.__name__, L)
            be.close()


def test_stager_multi_chunk_and_npy(mpc, oracle, configs, traces, tmp_path):
    # more than two 64 MiB staging chunks, plus the .npy streaming path with the
    # reference's dropped last row (LoaderNPY.cpp:28-32 + main.cpp:240)
    cfg = configs.probe_config(64)
    ev = mpc.VPC(cfg)
    n = (150 << 20) // 64 + 12345
    lines = traces.mixed(n, 64)
    lines[::7] = traces.random_u32((n + 6) // 7, 64)
    s, sel = ev.compress_lines(lines)
    o = oracle.VpcOracle(cfg)
    idx = np.concatenate([np.arange(0, 5000), np.arange(n // 2, n // 2 + 5000), np.arange(n - 5000, n)])
    sr, cr = o.compress(lines[idx])
    assert (s[idx] == sr).all() and (sel[idx] == cr).all()
    v = ev.stats_vector()
    assert v[0] == n and v[2] == int(s.astype(np.int64).sum())
    p = str(tmp_path / "trace.npy")
    traces.save_npy(p, lines[:200001])
    ev.reset()
    assert ev.compress_npy(p) == 200000
    o.reset(); o.compress(lines[:200000])
    assert (ev.stats_vector() == o.stats_vector()).all()
    ev.reset()
    assert ev.compress_npy(p, first_row=1000, n_rows=5000, skip_last_row=False) == 5000
    o.reset(); o.compress(lines[1000:6000])
    assert (ev.stats_vector() == o.stats_vector()).all()


def test_gpgpusim_log_streaming(mpc, oracle, configs, traces, tmp_path):
    """mpc_compress_gpgpusim_log: several staging chunks, every request type, an incomplete
    trailing request; against the oracle on oracle/gpgpusim_log.py's reading of the file."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    from oracle import gpgpusim_log as G
    L = 128
    n = 1_600_000                          # > two 64 MiB staging chunks of evaluated lines
    lines = traces.structured(n, L, seed=77)
    types = np.random.default_rng(5).choice([0, 4, 0, 4, 0, 4, 0, 1, 2, 8], n)
    one = open(traces.write_gpgpusim_log(str(tmp_path / "one.log"), lines[:1]), "rb").read()[1 + 7 * 17:]
    p = traces.write_gpgpusim_log(str(tmp_path / "big.log"), lines, types, tail=one[:-1])
    kept = lines[(types == 0) | (types == 4)]
    assert len(kept) > 2 * ((64 << 20) // L)
    assert mpc.gpgpusim_log_line_size(p) == L
    assert (G.evaluated_lines(p) == kept).all()
    # BDI against the oracle on everything; VPC against the (oracle-checked) host-buffer path on
    # everything and against the oracle on a prefix
    ev, o = mpc.BDI(L), oracle.BdiOracle(L)
    assert ev.compress_gpgpusim_log(p) == (n, len(kept))
    o.compress(kept)
    assert (ev.stats_vector() == o.stats_vector()).all()
    ev.close()
    cfg = configs.probe_config(L)
    ev, ev2 = mpc.VPC(cfg), mpc.VPC(cfg)
    assert ev.compress_gpgpusim_log(p) == (n, len(kept))
    ev2.compress_lines(kept)
    assert (ev.stats_vector() == ev2.stats_vector()).all()
    ev.close()
    ev2.close()
    small = traces.write_gpgpusim_log(str(tmp_path / "small.log"), lines[:30000], types[:30000])
    ev, o = mpc.VPC(cfg), oracle.VpcOracle(cfg)
    ev.compress_gpgpusim_log(small)
    o.compress(G.evaluated_lines(small))
    assert (ev.stats_vector() == o.stats_vector()).all()
    ev.close()
    # wrong line size / mixed sizes are errors, not crashes
    ev = mpc.BDI(64)
    with pytest.raises(mpc.MpcError) as e:
        ev.compress_gpgpusim_log(p)
    assert e.value.code == -22
    ev.close()


def test_device_resident_path_and_synth(mpc, oracle, configs, traces):
    import torch
    dev = torch.device("cuda:0")
    for kind, gen, L in (("random_u32", traces.random_u32, 64), ("sine_f32", traces.sine_f32, 64),
                         ("mixed", traces.mixed, 64), ("pointers_u64", traces.pointers_u64, 128),
                         ("zeros", traces.zeros, 32)):
        n, first = 40000, 123457
        buf = torch.empty(n * L, dtype=torch.uint8, device=dev)
        mpc.synth_fill(buf.data_ptr(), n, L, kind, first_line=first)
        torch.cuda.synchronize()
        host = buf.cpu().numpy().reshape(n, L)
        ref = gen(n, L) if kind == "zeros" else gen(n, L, first_line=first)
        assert (host == ref).all(), kind
        cfg = configs.probe_config(L)
        ev = mpc.VPC(cfg)
        d_s = torch.empty(n, dtype=torch.int16, device=dev)
        d_c = torch.empty(n, dtype=torch.int8, device=dev)
        ev.compress_device(buf.data_ptr(), n, d_s.data_ptr(), d_c.data_ptr(),
                           stream=torch.cuda.current_stream().cuda_stream)
        ev.sync(); torch.cuda.synchronize()
        o = oracle.VpcOracle(cfg)
        sr, cr = o.compress(ref)
        assert (d_s.cpu().numpy().view(np.uint16) == sr).all() and (d_c.cpu().numpy() == cr).all()
        assert (ev.stats_vector() == o.stats_vector()).all()
        # device-side exchange operand: raw accumulators copied on the stream, derived on the host
        scratch = torch.zeros(ev.stats_raw_len(), dtype=torch.int64, device=dev)
        ev.stats_copy_raw_device(scratch.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        raw = scratch.cpu().numpy().view(np.uint64)
        assert (ev.stats_from_raw(raw) == o.stats_vector()).all()
        assert (ev.stats_from_raw(raw * np.uint64(3))[:3] == o.stats_vector()[:3] * np.uint64(3)).all()   # sums of ranks
    ev = mpc.BDI(64)
    lines = traces.bdi_stress(2800, 64)
    ev.compress_lines(lines)
    scratch = torch.zeros(ev.stats_raw_len(), dtype=torch.int64, device=dev)
    ev.stats_copy_raw_device(scratch.data_ptr(), 0)
    torch.cuda.synchronize()
    assert (ev.stats_from_raw(scratch.cpu().numpy().view(np.uint64)) == ev.stats_vector()).all()


@pytest.mark.parametrize("L", [32, 64, 128])
def test_block_and_group_boundaries(mpc, oracle, configs, traces, L):
    """The lane kernel streams whole 128-line blocks through its rings and hands what is left behind the last whole block
    (fewer than 128 lines) to the drain copy of the group code; the BDI kernel streams whole groups of 64 lines through
    its ring and takes the last partial group from plain loads.  Every line count around those boundaries -- 1 line, one
    short of / exactly / one past a group and a block, several blocks -- device-resident and through the host stager,
    per-line results and statistics against the oracle."""
    import torch
    rng = np.random.default_rng(L)
    pool = np.concatenate([traces.structured(700, L, seed=5), traces.mixed(300, L), traces.random_u32(200, L),
                           traces.zeros(30, L), traces.word_same(30, L)])
    pool = pool[rng.permutation(len(pool))]
    cfg = configs.probe_config(L)
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 640, 1023, 1024, 1025):
        lines = pool[:n]
        d_lines = torch.from_numpy(np.ascontiguousarray(lines)).to("cuda:0")
        for make_ev, make_o in ((lambda: mpc.VPC(cfg), lambda: oracle.VpcOracle(cfg)), (lambda: mpc.BDI(L), lambda: oracle.BdiOracle(L))):
            ev, o = make_ev(), make_o()
            r = o.compress(lines)
            s_ref, k_ref = r if isinstance(r, tuple) else (r, None)
            d_s = torch.zeros(n, dtype=torch.int16, device="cuda:0")
            d_k = torch.zeros(n, dtype=torch.int8, device="cuda:0")
            ev.compress_device(d_lines.data_ptr(), n, d_s.data_ptr(), d_k.data_ptr())
            ev.sync()
            assert (d_s.cpu().numpy().view(np.uint16) == s_ref).all(), (L, n)
            if k_ref is not None:
                assert (d_k.cpu().numpy() == k_ref).all(), (L, n)
            assert (ev.stats_vector() == o.stats_vector()).all(), (L, n)
            # statistics-only launch (another instantiation of the kernel) on top: every counter doubles
            v1 = ev.stats_vector()
            ev.compress_device(d_lines.data_ptr(), n)
            assert (ev.stats_vector() == 2 * v1).all(), (L, n)
            ev.close()


def test_full_size_properties(mpc, configs, golden_dir):
    """Checks at a BASELINE-scale buffer: per-line outputs of 32 Mi random lines
    against the oracle-made golden list of the lines that do compress; counts add
    up; sharding the buffer and merging the statistics equals one pass; repeating
    a pass doubles every counter."""
    import json
    import os
    import torch
    dev = torch.device("cuda:0")
    L, n = 64, 32 << 20                      # 2 GiB of lines
    buf = torch.empty(n * L, dtype=torch.uint8, device=dev)
    cfg = configs.probe_config(L)
    with open(os.path.join(golden_dir, "random_u32_32Mi_exceptions.json")) as f:
        gold = json.load(f)
    assert gold["n_lines"] == n
    for kind in ("random_u32", "mixed"):
        mpc.synth_fill(buf.data_ptr(), n, L, kind)
        whole = mpc.VPC(cfg)
        d_s = torch.empty(n, dtype=torch.int16, device=dev)
        d_c = torch.empty(n, dtype=torch.int8, device=dev)
        whole.compress_device(buf.data_ptr(), n, d_s.data_ptr(), d_c.data_ptr())
        v = whole.stats_vector()
        assert v[0] == n and v[1] == n * 512
        K, B = 7, whole.hist_bins
        hist = v[3 + 6 * K:].reshape(K, B)
        assert int(hist.sum()) == n
        sizes = np.arange(B, dtype=np.uint64)
        assert int((hist * sizes[None, :]).sum()) == int(v[2])
        assert int(d_s.to(torch.int64).sum().item()) == int(v[2])
        if kind == "random_u32":
            # every line is 512 + 3 id bits / cluster -1, except the oracle's golden exceptions
            odd = torch.nonzero((d_s != gold["default_size"]) | (d_c != gold["default_cluster"])).flatten().cpu().numpy()
            got = [{"line": int(i), "size": int(d_s[int(i)].item()), "cluster": int(d_c[int(i)].item())} for i in odd]
            assert got == gold["exceptions"]
        # 3 unequal shards on separate handles, merged == whole
        parts = mpc.VPC(cfg)
        cuts = [0, n // 3 + 5, n // 2 + 77, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            sh = mpc.VPC(cfg)
            sh.compress_device(buf.data_ptr() + a * L, b - a)
            parts.stats_merge(sh.stats_vector())
            sh.close()
        assert (parts.stats_vector() == v).all()
        whole.compress_device(buf.data_ptr(), n)
        assert (whole.stats_vector() == 2 * v).all()
        whole.close(); parts.close()


def _gpu_shard_worker(rank, world, port, npy_path, out_dir):
    import importlib
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share the one GPU of the box
    mpc = importlib.import_module("cal_22-mpc_amd")
    sharded = importlib.import_module("cal_22-mpc_amd.sharded")
    configs = importlib.import_module("cal_22-mpc_amd.configs")
    ev = mpc.VPC(configs.probe_config(64), device=0)
    total = sharded.evaluate_sharded(ev, npy_path, rank, world)
    assert (ev.stats_vector() == total).all()
    np.save(os.path.join(out_dir, f"tot_{rank}.npy"), total)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_npy(mpc, oracle, configs, traces, tmp_path):
    """Two processes, each evaluating its contiguous half of one .npy trace on the
    GPU, one all-reduce of the statistics vector: equals the oracle over the
    whole trace (minus the reference's dropped last row)."""
    import socket
    import torch.multiprocessing as mp
    lines = np.concatenate([traces.structured(30000, 64, seed=21), traces.mixed(20001, 64)])
    p = traces.save_npy(str(tmp_path / "t.npy"), lines)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gpu_shard_worker, args=(2, port, p, str(tmp_path)), nprocs=2, join=True)
    o = oracle.VpcOracle(configs.probe_config(64))
    o.compress(lines[:-1])
    for r in range(2):
        assert (np.load(tmp_path / f"tot_{r}.npy") == o.stats_vector()).all()


def test_bench_two_rank_rehearsal_child_process(tmp_path):
    """bench.py's N > 1 path (contiguous shards, the all-reduce of the device accumulators, the
    max-over-ranks timing, the config 4 sub-record) run as a FRESH child process under
    torch.distributed.run -- two ranks on the one GPU of this box with the gloo backend (an RCCL group
    needs one GPU per rank).  The line-count assertions inside bench.py must hold and the JSON line
    must carry the N > 1 fields."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), "--log-dir", str(tmp_path / "elastic"),      # (else /tmp/torchelastic_*)
           os.path.join(root, "bench.py"),
           "--gpus", "2", "--rehearse-single-gpu", "--lines", "1048576", "--steps", "2", "--warmup", "1",
           # a config-4 shard LARGER than the primary buffer: the primary buffer is released and the shard allocated anew
           "--config4-lines", "3145728"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [l for l in p.stdout.split("\n") if l.startswith("{")]
    assert len(line) == 1, p.stdout[-2000:]
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["blocks_per_gpu"] == 1048576 and d["config"]["sharding"] == "contiguous x2"
    assert d["value"] > 0 and d["roofline"]["kernel"] == "vpc_lane_kernel<16>"
    # the collective's group size and backend at the top level of the line (a driver-run record can be checked for
    # "the collective saw N ranks" without parsing the sub-record)
    assert d["rccl_ranks"] == 2 and d["collective_backend"].startswith("gloo")
    c4 = d["config4"]
    assert c4["rccl_ranks"] == 2 and c4["sharding"] == "contiguous x2"
    assert c4["workload"].startswith(str(2 * 3145728) + " mixed")
    assert 0.0 <= c4["all_reduce_share_of_pass"] < 1.0 and c4["blocks_per_s"] > 0
    # random u32 blocks stay uncompressed at 515 bits, mixed blocks compress
    assert abs(d["config"]["compression_ratio"] - 512.0 / 515.0) < 1e-6
    assert c4["compression_ratio"] > 1.0


ROUTE_NAMES = ["vpc_deferred", "vpc_drains", "vpc_paired_blocks", "vpc_plain_blocks", "vpc_to_paired", "vpc_to_plain",
               "vpc_tail_groups", "bdi_deferred", "bdi_drains"]        # MPC_RT_* of csrc/mpc_kernel_common.h

ROUTES_PRELUDE = r"""
import ctypes, importlib, json, sys
import numpy as np
sys.path.insert(0, %r)
mpc = importlib.import_module("cal_22-mpc_amd"); C = importlib.import_module("cal_22-mpc_amd.configs"); T = importlib.import_module("cal_22-mpc_amd.traces")
from oracle import oracle as O
assert mpc.LIB_PATH.endswith("libmpc_hip_test.so"), mpc.LIB_PATH
def routes(ev):
    # test library only: how often the kernels' alternative routes ran since the statistics were last reset
    f = mpc.lib().mpc_test_routes
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]; f.restype = ctypes.c_int
    out = (ctypes.c_uint64 * 16)()
    assert f(ev._h, out, 16) == 0
    return dict(zip(%r, [int(x) for x in out]))
"""


def _run_with_test_library(code: str, grid_cap: int, timeout: int = 900):
    """A fresh process bound to libmpc_hip_test.so (the product library has neither route counters nor the grid
    cap), MPC_TEST_GRID set; returns the JSON object the code printed behind 'ROUTES '."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    test_lib = os.path.join(root, "cal_22-mpc_amd", "libmpc_hip_test.so")
    assert os.path.exists(test_lib), "libmpc_hip_test.so is missing: python cal_22-mpc_amd/build.py"
    env = dict(os.environ, MPC_TEST_GRID=str(grid_cap), MPC_HIP_LIB=test_lib)
    r = subprocess.run([sys.executable, "-c", (ROUTES_PRELUDE % (root, ROUTE_NAMES)) + code], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.split("\n") if l.startswith("ROUTES ")]
    assert line, r.stdout[-2000:]
    return json.loads(line[-1][7:])


def test_deferred_line_queues_fill_and_drain_inside_the_loop(oracle, configs, traces, tmp_path):
    """The VPC and BDI kernels set lines aside into per-wave LDS queues (a few hundred entries) and drain them when
    the queue is nearly full.  With the grid capped (MPC_TEST_GRID, test library only) every wave walks hundreds of
    groups of lines, so the queues fill and drain many times inside the loop; results against the oracle, in a fresh
    process (the cap is read once per process).  The test library's route counters say that lines really were set
    aside and that drains really ran inside the loop, not only at its end."""
    code = r"""
lines = np.concatenate([T.mixed(90000, 64), T.structured(60000, 64, seed=3), T.sine_f32(20000, 64), T.bdi_screen_stress(48000, 64),
                        T.random_u32(30001, 64)])
lines = lines[np.random.default_rng(7).permutation(len(lines))]
res = {}
for name, cfg in (("probe", C.probe_config(64)), ("mpc", C.mpc_config(64))):
    ev, o = mpc.VPC(cfg), O.VpcOracle(cfg)
    s, k = ev.compress_lines(lines)
    s_ref, k_ref = o.compress(lines)
    assert (s == s_ref).all() and (k == k_ref).all() and (ev.stats_vector() == o.stats_vector()).all()
    res[name] = routes(ev)
ev, o = mpc.BDI(64), O.BdiOracle(64)
s, k = ev.compress_lines(lines)
s_ref, k_ref = o.compress(lines)
assert (s == s_ref).all() and (k == k_ref).all() and (ev.stats_vector() == o.stats_vector()).all()
res["bdi"] = routes(ev)
print("ROUTES " + json.dumps(res))
"""
    res = _run_with_test_library(code, grid_cap=2)          # one large workgroup of the VPC lane kernel (16 waves); 8 waves for BDI
    for name in ("probe", "mpc"):
        r = res[name]
        # a wave's end-of-trace drain is at most 256 / 64 = 4 groups: more than 16 x 4 drain groups means that queues
        # were drained inside the streaming loop
        assert r["vpc_deferred"] > 0 and r["vpc_drains"] > 64, (name, r)
        assert r["vpc_deferred"] <= 64 * r["vpc_drains"], (name, r)
    b = res["bdi"]
    # 8 waves, 512 entries each: more than 8 x 8 drain groups cannot come from the end of the trace alone
    assert b["bdi_deferred"] > 0 and b["bdi_drains"] > 64 and b["bdi_deferred"] <= 64 * b["bdi_drains"], b


def test_general_layout_twins_under_a_capped_grid(oracle, configs, traces, tmp_path):
    """The general-layout twins (RootIndex 1..15, whole-plane truncation) share the streaming loop of the plain unrolled
    kernels: with the grid capped to one workgroup (test library) every wave reuses its ring stages hundreds of times,
    sets lines aside and drains them inside the loop, and switches to paired groups on the alternating trace -- all of it
    in the twins' copy of the group code.  Results against the oracle; the route counters say the routes ran."""
    code = r"""
res = {}
az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}
for L in (64, 32, 128):
    def trunc(ts): return None if ts is None else {"TableSize": ts, "Rows": [i // L for i in range(ts)], "Cols": [i % L for i in range(ts)]}
    prev4 = [max(i - 4, 0) for i in range(L)]; w2 = [[1.0, 0.5][i % 2] for i in range(L)]; d1 = [1 if i % 4 == 0 else 0 for i in range(L)]
    n = 120001 if L == 64 else 40003
    lines = np.concatenate([T.mixed(n, L), T.structured(n // 3, L, seed=3), T.sine_f32(n // 6, L), T.random_u32(n // 5, L), T.mixed(4097, L)])
    for roots, ts in (((5, 3, 2), None), ((0, 0, 0), 6 * L), ((13, 1, 2), 5 * L)):
        s_ = trunc(ts)
        cfg = C.make_config(L, [az, aws, C.one_base(L, roots[0], True, s_), C.consecutive_base(L, 0, True, s_), C.diff_base(L, prev4, d1, roots[1], False, s_),
                                C.weight_base(L, prev4, w2, roots[2], True, s_)])
        d = mpc.describe_config(cfg)
        assert d["sequence"] == "unrolled" and d["general_layout"] == "yes", d
        ev, o = mpc.VPC(cfg), O.VpcOracle(cfg)
        s, k = ev.compress_lines(lines)
        s_ref, k_ref = o.compress(lines)
        assert (s == s_ref).all() and (k == k_ref).all() and (ev.stats_vector() == o.stats_vector()).all(), (L, roots, ts)
        res["%d/%s/%s" % (L, roots, ts)] = routes(ev)
        ev.close()
# the byte-major order on kernels compiled at creation (the test library's own code object: its route counters are in)
L = 64
bm = {"TableSize": 8 * L, "Rows": [i % 8 for i in range(8 * L)], "Cols": [i // 8 for i in range(8 * L)]}
prev4 = [max(i - 4, 0) for i in range(L)]; w2 = [[1.0, 0.5][i % 2] for i in range(L)]; d1 = [1 if i % 4 == 0 else 0 for i in range(L)]
cfg = C.make_config(L, [az, aws, C.one_base(L, 0, True, bm), C.consecutive_base(L, 0, True, bm), C.diff_base(L, prev4, d1, 0, False, bm),
                        C.weight_base(L, prev4, w2, 0, True, bm)])
lines = np.concatenate([T.mixed(90001, L), T.structured(30000, L, seed=3), T.random_u32(20000, L), T.mixed(4097, L)])
ev, o = mpc.VPC(cfg), O.VpcOracle(cfg)
assert ev.kernel_form.startswith("unrolled, compiled at creation"), ev.kernel_form
s, k = ev.compress_lines(lines)
s_ref, k_ref = o.compress(lines)
assert (s == s_ref).all() and (k == k_ref).all() and (ev.stats_vector() == o.stats_vector()).all()
res["64/byte-major"] = routes(ev)
ev.close()
print("ROUTES " + json.dumps(res))
"""
    res = _run_with_test_library(code, grid_cap=1)
    for key, r in res.items():
        assert r["vpc_paired_blocks"] + r["vpc_plain_blocks"] > 0, (key, r)
        if key.startswith("64/") or key.startswith("32/"):           # (128-byte lines never set lines aside)
            assert r["vpc_deferred"] > 0 and r["vpc_drains"] > 0, (key, r)
    assert any(r["vpc_paired_blocks"] > 0 for r in res.values()), res


def test_paired_groups_on_alternating_lines(oracle, configs, traces, tmp_path):
    """Where neighbouring lines alternate between two kinds the VPC lane kernel switches a wave to paired groups
    (even lines of a 128-line block, then the odd ones) and probes with plain groups now and then.  With the grid
    capped to one workgroup (MPC_TEST_GRID, test library only) every wave walks hundreds of blocks, so it switches back
    and forth; traces that alternate throughout, that alternate in stretches between other data, and that end inside
    a block; per-line results and statistics against the oracle, in a fresh process (the cap is read once per
    process).  The test library's route counters say that paired blocks were evaluated, that waves went paired ->
    plain (probe) -> paired again, and that non-alternating data stayed plain."""
    code = r"""
res = {}
for L, n in ((64, 150001), (32, 60037), (128, 50003)):
    alt = T.mixed(n, L)
    parts = [T.mixed(20000 + 77, L), T.random_u32(9000 + 5, L), T.mixed(30001, L, first_line=1), T.structured(12000, L, seed=3),
             T.mixed(8192, L), T.zeros(300, L), T.mixed(4097, L)]
    for tname, lines in (("alt", alt), ("parts", np.concatenate(parts)), ("random", T.random_u32(40000, L))):
        for cname, cfg in (("probe", C.probe_config(L)), ("mpc", C.mpc_config(L))):
            ev, o = mpc.VPC(cfg), O.VpcOracle(cfg)
            s, k = ev.compress_lines(lines)
            s_ref, k_ref = o.compress(lines)
            assert (s == s_ref).all() and (k == k_ref).all() and (ev.stats_vector() == o.stats_vector()).all(), (L, len(lines))
            res["%d/%s/%s" % (L, tname, cname)] = routes(ev)
            ev.close()
# the reference-derived whole-line vectors of the interleaved trace (tests/golden/ref_line_vectors.json: numbers of
# the reference's compiled stage classes), now evaluated in paired groups
sys.path.insert(0, "tests")
import ref_lines
n_ref = 0
for name, cfg, lines, c in ref_lines.load_cases():
    if not name.startswith("mixed/"):
        continue
    size, sel = ref_lines.expected_sizes(O, cfg, lines, c)
    ev = mpc.VPC(cfg)
    s, k = ev.compress_lines(lines)
    assert (s == size).all() and (k == sel).all(), name
    ev.close()
    n_ref += 1
assert n_ref >= 2
print("ROUTES " + json.dumps(res))
"""
    res = _run_with_test_library(code, grid_cap=1)
    for key, r in res.items():
        L, tname, cname = key.split("/")
        blocks = r["vpc_paired_blocks"] + r["vpc_plain_blocks"]
        if cname == "mpc" and blocks == 0:
            continue                  # (a configuration that runs the run-time module loop: no ring, no blocks)
        assert blocks > 0, (key, r)
        if tname == "random":
            assert r["vpc_paired_blocks"] == 0 and r["vpc_to_paired"] == 0, (key, r)
        elif cname == "probe":
            # the probe configuration's modules tell the two kinds of line apart: most blocks of the alternating trace
            # are paired, and every wave that stayed long enough probed with plain groups and came back
            assert r["vpc_paired_blocks"] > 0 and r["vpc_to_paired"] > 0, (key, r)
            if tname == "alt" and L != "128":
                assert r["vpc_paired_blocks"] > r["vpc_plain_blocks"], (key, r)
            if tname == "alt" and L == "64":
                # 1171 blocks over 16 waves: every wave passes the probe interval (64 paired blocks) once,
                # goes back to plain groups and returns to paired ones
                assert r["vpc_to_plain"] >= 8 and r["vpc_to_paired"] > r["vpc_to_plain"], (key, r)
