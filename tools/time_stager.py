#!/usr/bin/env python3
"""Development helper: PCIe-inclusive rates of the staged paths (never the headline value):
host buffer -> mpc_compress_batch, .npy file -> mpc_compress_npy, .log file -> mpc_compress_gpgpusim_log."""
import importlib, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
mpc = importlib.import_module("cal_22-mpc_amd"); cfgs = importlib.import_module("cal_22-mpc_amd.configs")
traces = importlib.import_module("cal_22-mpc_amd.traces")
L = 64
n = (4 << 30) // L
rng = np.random.default_rng(1)
lines = rng.integers(0, 256, (n, L), dtype=np.uint8)
ev = mpc.VPC(cfgs.probe_config(L))
ev.compress_lines(lines[:1 << 20], False, False)           # warm-up (allocates the pinned slots)
for what in ("statistics only", "with per-line outputs"):
    t0 = time.perf_counter()
    ev.compress_lines(lines, what != "statistics only", what != "statistics only")
    ev.sync()
    dt = time.perf_counter() - t0
    print(f"host buffer, {what}: {n * L / dt / 1e9:.1f} GB/s ({n / dt / 1e9:.2f} G blocks/s)")
d = tempfile.mkdtemp(dir="/tmp")
p = traces.save_npy(os.path.join(d, "t.npy"), lines[: n // 2])
for rep in range(2):
    t0 = time.perf_counter()
    rows = ev.compress_npy(p)
    dt = time.perf_counter() - t0
    print(f".npy file (page cache{' warm' if rep else ''}): {rows * L / dt / 1e9:.1f} GB/s")
q = traces.write_gpgpusim_log(os.path.join(d, "t.log"), lines[: n // 4])
for rep in range(2):
    t0 = time.perf_counter()
    req, done = ev.compress_gpgpusim_log(q)
    dt = time.perf_counter() - t0
    print(f".log file (page cache{' warm' if rep else ''}): {done * L / dt / 1e9:.1f} GB/s of line data, {req / dt / 1e6:.1f} M requests/s")
os.remove(p); os.remove(q); os.rmdir(d)
