"""CPU restatement of the reference's GPGPU-Sim ".log" ingestion -- TEST INFRASTRUCTURE
ONLY (like oracle/mpc_oracle.c): tests compare the product's C++ loader
(cal_22-mpc_amd/host/LoaderGPGPU.cpp) and CLI against it.

Follows trace::gpgpusim::LoaderGPGPU (reference src/loader/LoaderGPGPU.cpp):
  * isFileValid (:93-119): first byte must be NUM_KEYS = 17, then 17 x (6-byte key
    name + 1-byte size), not interpreted;
  * GetCacheline (:26-55): 62-byte record header, field by field, then req_size data
    bytes; the read that runs into the end of the file raises isEnd, and the driver
    tests isEnd before using the record, so an incomplete trailing record is dropped
    and every complete record is delivered;
  * GetCachelineSize (:16-24): req_size of the first record;
and the driver's filter (reference src/main.cpp:222-224): only GLOBAL_ACC_R (0) and
GLOBAL_ACC_W (4) requests are evaluated.

Parity is pinned by the format as read from the source only: the reference ships no
.log fixture ("loader parity unpinned" in DESIGN.md)."""
import numpy as np

NUM_KEYS = 17
HEADER_BYTES = 1 + 7 * NUM_KEYS
RECORD_HEADER = 62


def read_records(path):
    """All complete records: list of (req_type, req_size, data bytes)."""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size < 1 or raw[0] != NUM_KEYS or raw.size < HEADER_BYTES:
        raise ValueError("The header of the GPGPU-sim trace file is not valid.")
    out = []
    p = HEADER_BYTES
    while p + RECORD_HEADER <= raw.size:
        req_type = int(raw[p + 38:p + 42].view("<u4")[0])
        req_size = int(raw[p + 58:p + 62].view("<u4")[0])
        q = p + RECORD_HEADER + req_size
        if q > raw.size:
            break
        out.append((req_type, req_size, raw[p + RECORD_HEADER:q]))
        p = q
    return out


def line_size(path) -> int:
    recs = read_records(path)
    return recs[0][1] if recs else 0


def evaluated_lines(path) -> np.ndarray:
    """The lines the reference driver hands to CompressLine, as a uint8 [n, L] array."""
    recs = read_records(path)
    L = recs[0][1] if recs else 0
    keep = [d for (t, sz, d) in recs if t in (0, 4)]
    if any(d.size != L for d in keep):
        raise ValueError("mixed request sizes")
    return np.stack(keep).astype(np.uint8) if keep else np.zeros((0, L), dtype=np.uint8)
