"""CPU restatement of the reference's APSim ".txt" ingestion -- TEST INFRASTRUCTURE ONLY (like
oracle/gpgpusim_log.py): tests compare the product's C++ loader (cal_22-mpc_amd/host/LoaderAPSim.cpp)
and CLI against it.

Follows trace::apsim::LoaderGPGPU (reference src/loader/LoaderGPGPU.cpp:116-525):
  * isFileValid (:469-510): first row = column names; a column containing "last" -> read trace, one
    containing "strb" -> write trace (first match decides), neither -> error;
  * readLineR / readLineW (:177-228): std::getline, then eof() -> end (a last row without a newline
    is dropped); 18 comma-separated fields: cycle, clock, valid0..3, data0..3 (64 hex digits = 32
    bytes, byte j = digits 2j, 2j+1), ready0..3, last0..3 | strb0..3;
  * getCacheline32 (:230-288): rows with clock == 0 or without a channel whose valid == 1 and
    ready == 1 are skipped; every handshaking channel of a row, in channel order, yields its 32 bytes;
  * getCacheline64 (:330-441): two consecutive beats of one channel make a 64-byte line (first beat
    first); beats without a partner at the end of the file are dropped.

Parity is pinned by the source reading only: the reference ships no .txt fixture."""
import numpy as np

NUM_CH = 4
ACCESS_GRAN = 32


def read_rows(path):
    raw = open(path, "rb").read().decode("latin-1")
    rows = raw.split("\n")
    # std::getline + eof(): only rows terminated by a newline count
    complete = rows[:-1]
    if not complete:
        raise ValueError("The header of the GPU traffic file is not valid.")
    header = complete[0].split(",")
    rw = None
    for col in header:
        if "last" in col:
            rw = "R"
            break
        if "strb" in col:
            rw = "W"
            break
    if rw is None:
        raise ValueError("The header of the GPU traffic file is not valid.")
    out = []
    for r in complete[1:]:
        f = r.split(",")
        out.append(dict(cycle=int(f[0]), clock=int(f[1]) & 0xff, valid=[int(x) & 0xff for x in f[2:6]],
                        data=[bytes.fromhex(x[:2 * ACCESS_GRAN]) for x in f[6:10]], ready=[int(x) & 0xff for x in f[10:14]]))
    return rw, out


def lines(path, line_size: int = 32) -> np.ndarray:
    """The lines the reference driver hands to CompressLine, as a uint8 [n, line_size] array."""
    assert line_size in (32, 64)
    _, rows = read_rows(path)
    out, pending = [], [[] for _ in range(NUM_CH)]
    for r in rows:
        if not r["clock"] or not any(r["valid"]):
            continue
        chans = [c for c in range(NUM_CH) if r["valid"][c] == 1 and r["ready"][c] == 1]
        for c in chans:
            if line_size == 32:
                out.append(r["data"][c])
            else:
                pending[c].append(r["data"][c])
        if line_size == 64:
            for c in range(NUM_CH):
                if len(pending[c]) == 2:
                    out.append(pending[c][0] + pending[c][1])
                    pending[c] = []
    if not out:
        return np.zeros((0, line_size), dtype=np.uint8)
    return np.frombuffer(b"".join(out), dtype=np.uint8).reshape(-1, line_size).copy()
