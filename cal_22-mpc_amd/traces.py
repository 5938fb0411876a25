"""Synthetic trace generators (host side, numpy) for the workloads of
BASELINE.json / SURVEY.md 8d.  Each returns a C-order ``uint8`` array of shape
``[n_lines, line_size]`` -- the layout ``trace::LoaderNPY`` serves
(reference ``src/loader/LoaderNPY.cpp:14-54``).

The full-size device-resident versions used by ``bench.py`` are produced by the
HIP generators behind ``mpc_synth_*`` (``csrc/mpc_synth.hip``), which are checked
against these functions on a prefix (tests/test_gpu_parity.py).
"""
from __future__ import annotations

import numpy as np

SEED = 12345
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Counter-based generator: splitmix64 finaliser of (index + seed stream).
    Pure function of the counter, so any shard can be generated independently."""
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _rand_u32(first_word: int, n_words: int, seed: int = SEED) -> np.ndarray:
    idx = np.arange(first_word, first_word + n_words, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = idx + np.uint64(seed) * np.uint64(0xD1342543DE82EF95)
    return (splitmix64(key) >> np.uint64(32)).astype(np.uint32)


def zeros(n_lines: int, line_size: int = 64) -> np.ndarray:
    return np.zeros((n_lines, line_size), dtype=np.uint8)


def random_u32(n_lines: int, line_size: int = 64, first_line: int = 0, seed: int = SEED) -> np.ndarray:
    """Config 2: every 32-bit word i.i.d. uniform, little-endian."""
    wpl = line_size // 4
    w = _rand_u32(first_line * wpl, n_lines * wpl, seed)
    return w.astype("<u4").view(np.uint8).reshape(n_lines, line_size)


def sine_table() -> np.ndarray:
    """float32(sin(2*pi*t/1024)) for t in [0,1024): the period of config 3."""
    t = np.arange(1024, dtype=np.float64)
    return np.sin(2.0 * np.pi * t / 1024.0).astype(np.float32)


def sine_f32(n_lines: int, line_size: int = 64, first_line: int = 0) -> np.ndarray:
    """Config 3: word t (global index) = float32(sin(2*pi*t/1024))."""
    wpl = line_size // 4
    t = np.arange(first_line * wpl, (first_line + n_lines) * wpl, dtype=np.int64)
    w = sine_table()[t % 1024]
    return w.astype("<f4").view(np.uint8).reshape(n_lines, line_size)


def mixed(n_lines: int, line_size: int = 64, first_line: int = 0) -> np.ndarray:
    """Config 4: even lines = 16 x uint32((16*line + j) mod 1000); odd lines =
    the sine words of config 3 at the same global word index."""
    wpl = line_size // 4
    lines = np.arange(first_line, first_line + n_lines, dtype=np.int64)
    j = np.arange(wpl, dtype=np.int64)
    gidx = lines[:, None] * wpl + j[None, :]
    ints = ((lines[:, None] * 16 + j[None, :]) % 1000).astype("<u4")
    sines = sine_table()[gidx % 1024].astype("<f4").view("<u4")
    w = np.where((lines % 2 == 0)[:, None], ints, sines).astype("<u4")
    return np.ascontiguousarray(w).view(np.uint8).reshape(n_lines, line_size)


def pointers_u64(n_lines: int, line_size: int = 128, first_line: int = 0, seed: int = SEED) -> np.ndarray:
    """Config 5: qwords 0x00007f3a5c000000 + 8*u, u uniform in [0, 2^20)."""
    qpl = line_size // 8
    u = _rand_u32(first_line * qpl, n_lines * qpl, seed).astype(np.uint64) & np.uint64((1 << 20) - 1)
    q = np.uint64(0x00007F3A5C000000) + np.uint64(8) * u
    return q.astype("<u8").view(np.uint8).reshape(n_lines, line_size)


def word_same(n_lines: int, line_size: int = 64, seed: int = SEED) -> np.ndarray:
    """Every line repeats one random non-zero 32-bit word."""
    w = _rand_u32(0, n_lines, seed) | np.uint32(1)
    return np.repeat(w.astype("<u4"), line_size // 4).view(np.uint8).reshape(n_lines, line_size)


def counters_u32(n_lines: int, line_size: int = 32, first_line: int = 0) -> np.ndarray:
    """Small incrementing 32-bit counters (a compressible integer pattern)."""
    wpl = line_size // 4
    idx = np.arange(first_line * wpl, (first_line + n_lines) * wpl, dtype=np.int64)
    return (idx % 251).astype("<u4").view(np.uint8).reshape(n_lines, line_size)


def bdi_screen_stress(n_lines: int, line_size: int = 64, seed: int = 4242) -> np.ndarray:
    """Lines around the thresholds of the BDI kernel's screening test: for a base
    size B the line is random B-byte values (every scan fails) with 0 .. T+2
    immediates for each delta size (T = n / (8 (B - D)), the largest count for
    which a failed scan still costs >= 8 L bits), immediates placed at value 0, at
    the screen's witness positions (values 1, n/2, n-1) or anywhere, negative /
    sign-extended 8-byte immediates, lines whose only misfitting delta lies at a
    value that is no witness, and deltas of exactly -1.  Interleaved with plain
    random lines so that waves hold both kinds."""
    rng = np.random.default_rng(seed)
    L = line_size
    out = rng.integers(0, 256, (n_lines, L), dtype=np.uint8)
    for i in range(n_lines):
        k = i % 8
        if k >= 5:
            continue                                   # plain random line
        B = (2, 4, 8)[(i // 8) % 3]
        n = L // B
        bits = 8 * B
        D = {2: (1,), 4: (1, 2), 8: (1, 2, 4)}[B]
        d_sz = D[(i // 24) % len(D)]
        T = n // (8 * (B - d_sz))
        vals = [int(x) for x in rng.integers(1 << (bits - 2), 1 << (bits - 1), n, dtype=np.uint64)]
        if k == 0 or k == 1:
            # c immediates for delta size d_sz at chosen positions
            c = int(rng.integers(0, T + 3))
            where = (i // 72) % 3
            pos = list(rng.permutation(n)[:c])
            if where == 0 and c:
                pos[0] = 0
            elif where == 1 and c:
                pos[0] = (1, n // 2, n - 1)[int(rng.integers(0, 3))]
            for p_ in pos:
                v = int(rng.integers(0, 1 << (8 * d_sz)))
                if B == 8 and k == 1:
                    v = (-int(rng.integers(2, 1 << (8 * d_sz - 1)))) % (1 << 64)   # negative immediate
                vals[int(p_)] = v
        elif k == 2:
            # all deltas fit except one at a value that is no witness
            base = vals[0]
            vals = [(base + int(x)) % (1 << bits) for x in rng.integers(0, 1 << (8 * d_sz - 1), n)]
            vals[0] = base
            free = [j for j in range(2, n - 1) if j != n // 2]
            if free:
                vals[free[int(rng.integers(0, len(free)))]] = int(rng.integers(1 << (bits - 2), 1 << (bits - 1), dtype=np.uint64))
        elif k == 3:
            # everything fits (the combination is selected), with a -1 delta in every other such line
            base = vals[0]
            vals = [(base + int(x)) % (1 << bits) for x in rng.integers(0, 1 << (8 * d_sz - 1), n)]
            vals[0] = base
            if (i // 8) % 2 and n > 1:
                vals[int(rng.integers(1, n))] = (base + 1) % (1 << bits)
        else:
            # value 0 immediate, the rest random (base is a later value)
            vals[0] = int(rng.integers(0, 1 << (8 * d_sz)))
        out[i] = np.array(vals, dtype=np.uint64).astype({2: "<u2", 4: "<u4", 8: "<u8"}[B]).view(np.uint8)
    return out


def bdi_stress(n_lines: int, line_size: int = 64, seed: int = 777) -> np.ndarray:
    """Lines that exercise every BDI mode (reference ``BDI.cpp:6-74``): zeros,
    8-byte repeats, each (base, delta) combination, deltas of exactly -1
    (rejected by reduceSign, ``BDI.cpp:203-218``), +128..+255 one-byte deltas
    (accepted), all-immediate lines (the unsigned wrap of ``BDI.cpp:200``) and
    random lines."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n_lines, line_size), dtype=np.uint8)
    for i in range(n_lines):
        k = i % 14
        if k == 0:
            continue
        if k == 1:
            q = rng.integers(1, 1 << 63, dtype=np.uint64)
            out[i] = np.full(line_size // 8, q, dtype="<u8").view(np.uint8)
            continue
        if k == 13:
            out[i] = rng.integers(0, 256, line_size, dtype=np.uint8)
            continue
        base_size, delta_size = [(8, 1), (8, 2), (8, 4), (4, 1), (4, 2), (2, 1),
                                 (8, 1), (4, 1), (2, 1), (8, 2), (4, 2)][k - 2]
        n = line_size // base_size
        bits = 8 * base_size
        base = int(rng.integers(1 << (bits - 2), 1 << (bits - 1), dtype=np.uint64))
        if k >= 8:
            variant = (i // 14) % 4
        else:
            variant = 0
        if variant == 0:      # small non-negative deltas
            d = rng.integers(0, 1 << (8 * delta_size - 1), n)
        elif variant == 1:    # a delta of exactly -1 somewhere
            d = rng.integers(0, 4, n)
            d[rng.integers(1, n)] = -1
        elif variant == 2:    # +128..+255-style deltas (top bit of the delta set)
            d = rng.integers(1 << (8 * delta_size - 1), 1 << (8 * delta_size), n)
        else:                 # all-immediate line (values fit the delta width)
            base = 0
            d = rng.integers(0, 1 << (8 * delta_size), n)
        vals = [(base + int(x)) % (1 << bits) for x in d]
        if variant != 3 and n > 2:
            vals[int(rng.integers(1, n))] = int(rng.integers(0, 1 << (8 * delta_size)))  # an immediate
        arr = np.array(vals, dtype=np.uint64)
        out[i] = arr.astype({2: "<u2", 4: "<u4", 8: "<u8"}[base_size]).view(np.uint8)
    return out


def save_npy(path: str, lines: np.ndarray) -> str:
    assert lines.dtype == np.uint8 and lines.ndim == 2
    np.save(path, np.ascontiguousarray(lines))
    return path


def structured(n_lines: int, line_size: int = 64, seed: int = 4242) -> np.ndarray:
    """A mix of compressible patterns that drives every branch of the selector and
    of the common encoder (zero runs, single ones, adjacent pairs, half rows)."""
    rng = np.random.default_rng(seed)
    L, W = line_size, line_size // 4
    out = np.zeros((n_lines, L), dtype=np.uint8)
    for i in range(n_lines):
        k = i % 12
        if k == 0:      # sparse bytes
            m = int(rng.integers(1, 6))
            out[i, rng.integers(0, L, m)] = rng.integers(1, 256, m)
        elif k == 1:    # repeated word + low-bit noise
            w = rng.integers(0, 1 << 32, dtype=np.uint64).astype(np.uint32)
            noise = rng.integers(0, 1 << int(rng.integers(1, 9)), W).astype(np.uint32)
            out[i] = (w + noise).astype("<u4").view(np.uint8)
        elif k == 2:    # counters with a random stride
            s0, st = int(rng.integers(0, 1 << 30)), int(rng.integers(0, 300))
            out[i] = (s0 + st * np.arange(W)).astype("<u4").view(np.uint8)
        elif k == 3:    # fp32 ramp
            a, b = rng.normal(), rng.normal() * 1e-3
            out[i] = (a + b * np.arange(W)).astype("<f4").view(np.uint8)
        elif k == 4:    # sine, random phase
            t = int(rng.integers(0, 1 << 20)) + np.arange(W)
            out[i] = np.sin(2 * np.pi * t / 1024).astype("<f4").view(np.uint8)
        elif k == 5:    # byte ramp
            out[i] = (int(rng.integers(0, 256)) + int(rng.integers(0, 4)) * np.arange(L)).astype(np.uint8)
        elif k == 6:    # one byte value everywhere but a few positions
            out[i] = int(rng.integers(1, 256))
            m = int(rng.integers(0, 4))
            out[i, rng.integers(0, L, m)] = rng.integers(0, 256, m)
        elif k == 7:    # single bit set somewhere
            out[i, int(rng.integers(0, L))] = 1 << int(rng.integers(0, 8))
        elif k == 8:    # two adjacent bytes with the same single bit
            p = int(rng.integers(0, L - 1))
            b = 1 << int(rng.integers(0, 8))
            out[i, p] = b
            out[i, p + 1] = b
        elif k == 9:    # 16-bit values
            out[i] = rng.integers(0, 1 << int(rng.integers(1, 17)), L // 2).astype("<u2").view(np.uint8)
        elif k == 10:   # pointer-like qwords
            u = rng.integers(0, 1 << 16, L // 8).astype(np.uint64)
            out[i] = (np.uint64(0x00007F3A5C000000) + np.uint64(8) * u).astype("<u8").view(np.uint8)
        else:           # random with zeroed upper bytes
            w = rng.integers(0, 1 << int(rng.integers(1, 33)), W, dtype=np.uint64).astype("<u4")
            out[i] = w.view(np.uint8)
    return out


# ---- GPGPU-Sim ".log" traces (reference src/loader/LoaderGPGPU.cpp:26-47, 93-119) -------------
GPGPUSIM_KEYS = [("kid", 1), ("mftype", 1), ("cycle", 8), ("tpc", 4), ("sid", 4), ("wid", 4), ("pc", 4),
                 ("instct", 4), ("addr", 8), ("reqtyp", 4), ("row", 4), ("chip", 4), ("bank", 4), ("col", 4),
                 ("reqsiz", 4), ("data", 0), ("pad", 0)]
GLOBAL_ACC_R, LOCAL_ACC_R, CONST_ACC_R, TEXTURE_ACC_R, GLOBAL_ACC_W, LOCAL_ACC_W, L1_WRBK_ACC, L2_WRBK_ACC, INST_ACC_R = range(9)


def write_gpgpusim_log(path: str, lines: np.ndarray, req_types=None, seed: int = 3, tail: bytes = b"") -> str:
    """Synthetic GPGPU-Sim memory-request trace in the layout the reference's loader
    reads: 1 byte key count (17), 17 x (6-byte key name, 1-byte size), then per
    request a 62-byte little-endian header (kid u8, mf_type u8, cycle u64, tpc, sid,
    wid, pc, inst_cnt u32, mem_addr u64, req_type, row, chip, bank, col, req_size u32)
    followed by req_size data bytes.  ``req_types[i]`` is the request type of line i
    (default: all GLOBAL_ACC_R); ``tail`` is appended verbatim (e.g. an incomplete
    record)."""
    lines = np.ascontiguousarray(lines, dtype=np.uint8)
    n, L = lines.shape
    rng = np.random.default_rng(seed)
    if req_types is None:
        req_types = np.zeros(n, dtype=np.uint32)
    req_types = np.asarray(req_types, dtype=np.uint32)
    rec = np.zeros((n, 62 + L), dtype=np.uint8)
    rec[:, 0] = rng.integers(0, 4, n)
    rec[:, 1] = np.where(req_types == GLOBAL_ACC_W, 1, 0)
    rec[:, 2:10] = np.cumsum(rng.integers(1, 50, n)).astype("<u8").view(np.uint8).reshape(n, 8)
    for off in (10, 14, 18, 22, 26):
        rec[:, off:off + 4] = rng.integers(0, 1 << 16, n).astype("<u4").view(np.uint8).reshape(n, 4)
    rec[:, 30:38] = (0x7f0000000000 + rng.integers(0, 1 << 30, n).astype(np.uint64) * L).astype("<u8").view(np.uint8).reshape(n, 8)
    rec[:, 38:42] = req_types.astype("<u4").view(np.uint8).reshape(n, 4)
    for off in (42, 46, 50, 54):
        rec[:, off:off + 4] = rng.integers(0, 1 << 10, n).astype("<u4").view(np.uint8).reshape(n, 4)
    rec[:, 58:62] = np.full(n, L, dtype="<u4").view(np.uint8).reshape(n, 4)
    rec[:, 62:] = lines
    with open(path, "wb") as f:
        f.write(bytes([len(GPGPUSIM_KEYS)]))
        for name, size in GPGPUSIM_KEYS:
            f.write(name.encode().ljust(6, b"\0")[:6] + bytes([size]))
        f.write(rec.tobytes())
        f.write(tail)
    return path



# ---- APSim ".txt" traces (reference src/loader/LoaderGPGPU.cpp:177-228, 469-510) ----------------
def write_apsim_txt(path: str, beats: np.ndarray, write_trace: bool = False, seed: int = 5, final_newline: bool = True,
                    idle_rows: bool = True) -> str:
    """Synthetic APSim traffic file in the layout the reference's loader reads: a header row naming the columns
    (with "last" columns for a read trace, "strb" for a write trace), then rows
    cycle,clock,valid0..3,data0..3,ready0..3,last0..3|strb0..3 with 32-byte data beats as 64 hex digits.
    ``beats`` ([n, 32] uint8) are dealt to random channels of handshaking rows (1-4 per row); rows with clock 0,
    rows without a valid channel and rows with valid but not ready channels (carrying other data) are mixed in."""
    beats = np.ascontiguousarray(beats, dtype=np.uint8)
    assert beats.ndim == 2 and beats.shape[1] == 32
    rng = np.random.default_rng(seed)
    tail = "strb" if write_trace else "last"
    cols = ["Time", "clk"] + [f"valid_{i}" for i in range(4)] + [f"data_{i}" for i in range(4)] + \
           [f"ready_{i}" for i in range(4)] + [f"{tail}_{i}" for i in range(4)]
    rows = [",".join(cols)]
    cycle, i, n = 1000, 0, len(beats)

    def row(clock, valid, ready, data):
        extra = [format(int(rng.integers(0, 1 << 32)), "08x") if write_trace else str(int(rng.integers(0, 2))) for _ in range(4)]
        return ",".join([str(cycle), str(clock)] + [str(v) for v in valid] + [bytes(d).hex() for d in data] +
                        [str(r) for r in ready] + extra)
    while i < n:
        cycle += int(rng.integers(1, 9))
        junk = rng.integers(0, 256, (4, 32), dtype=np.uint8)
        kind = int(rng.integers(0, 10)) if idle_rows else 9
        if kind == 0:        # falling clock edge: ignored whatever else it says
            rows.append(row(0, [1, 1, 1, 1], [1, 1, 1, 1], junk))
        elif kind == 1:      # nothing valid
            rows.append(row(1, [0, 0, 0, 0], [int(x) for x in rng.integers(0, 2, 4)], junk))
        elif kind == 2:      # valid without ready: no handshake
            v = [int(x) for x in rng.integers(0, 2, 4)]
            rows.append(row(1, v, [0 if x else 1 for x in v], junk))
        else:
            k = min(int(rng.integers(1, 5)), n - i)
            chans = sorted(rng.permutation(4)[:k].tolist())
            valid, ready, data = [0] * 4, [0] * 4, junk.copy()
            for c in chans:
                valid[c], ready[c] = 1, 1
                data[c] = beats[i]
                i += 1
            for c in range(4):      # other channels: valid or ready alone, never both
                if c not in chans:
                    valid[c], ready[c] = [(0, 0), (1, 0), (0, 1)][int(rng.integers(0, 3))]
            rows.append(row(1, valid, ready, data))
    with open(path, "w", newline="") as f:
        f.write("\n".join(rows) + ("\n" if final_newline else ""))
    return path
