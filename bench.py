#!/usr/bin/env python3
"""bench.py -- headline measurement: 64 B blocks/s through the VPC evaluator on
device-resident synthetic traces, with the kernel's achieved fraction of the
MI355X HBM read roofline and a timed CPU baseline beside it.

    python bench.py --gpus 1 --steps 20 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the evaluator over this rank's shard of the trace (one
kernel launch over `--lines` blocks already resident in HBM) followed, for N>1,
by the one real exchange of the path: a sum all-reduce (RCCL) of the integer
statistics vector.  Lines shard contiguously over ranks (weak scaling: the
per-GPU shard is fixed, config.workload names it); there is no other collective.

Prints ONE JSON line on rank 0.  Beside the headline fields (BASELINE configs[1]:
random-uint32 64 B blocks) the line carries, outside the timed region,
  "workloads": N = 1 -- the other BASELINE workloads (fp32 sine, mixed int/fp, zeros,
               128 B pointer qwords) through VPC and BDI: kernel, kernel_ms_avg, roofline, ratio;
  "config4":   N > 1 -- BASELINE configs[3] (mixed int/fp, 512 Mi blocks per GPU, sharded
               contiguously, one all-reduce per pass): rccl_ranks, pass time, all-reduce share.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "cal_22-mpc_amd"
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (synth kind, line size, config builder line size)
    "random_u32": ("random_u32", 64),
    "sine_f32": ("sine_f32", 64),
    "mixed": ("mixed", 64),
    "zeros": ("zeros", 64),
    "pointers_u64_128": ("pointers_u64", 128),
    # 32-byte blocks (the block size of the paper's figure, reference MPC.PNG; north_star: "32/64/128 B")
    "random_u32_32": ("random_u32", 32),
    "mixed_32": ("mixed", 32),
}


def cpu_baseline(configs, traces, workload: str, L: int, seconds: float = 12.0, algo: str = "VPC"):
    """The oracle (a CPU port of the reference algorithm, NOT the product path) timed on
    this host's cores over a bounded sample of the same workload."""
    from oracle import oracle as O
    kind = WORKLOADS[workload][0]
    gen = {"random_u32": traces.random_u32, "sine_f32": traces.sine_f32, "mixed": traces.mixed,
           "zeros": traces.zeros, "pointers_u64": traces.pointers_u64}[kind]
    host_cores = len(os.sched_getaffinity(0))      # cores this process may run on ...
    quota = None                                   # ... and the CPU time its control group grants (cores' worth)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    cores = max(1, host_cores if quota is None else min(host_cores, int(quota + 0.5)))     # one thread per usable core
    cfg = configs.probe_config(L)
    make = {"VPC": lambda: O.VpcOracle(cfg), "BDI": lambda: O.BdiOracle(L), "FPC": lambda: O.FpcOracle(L), "BPC": lambda: O.BpcOracle(L)}[algo]

    def run(per_thread):
        data = gen(per_thread, L)
        oracles = [make() for _ in range(cores)]
        ths = [threading.Thread(target=oracles[i].compress, args=(data,)) for i in range(cores)]  # ctypes drops the GIL
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return time.perf_counter() - t0
    # single-thread rate on a short run; then a pilot with every thread running (the threads may share fewer cores than
    # they are), which sizes the sample for ~`seconds` of wall time
    cal_n = 4096
    cal = gen(cal_n, L)
    o = make()
    t0 = time.perf_counter()
    o.compress(cal, stats=True)
    rate1 = cal_n / (time.perf_counter() - t0)
    pilot_n = 2048
    pilot_rate = cores * pilot_n / run(pilot_n)
    per_thread = int(min(max(pilot_rate * seconds / cores, 2048), 4 << 20))
    dt = run(per_thread)
    if dt < seconds / 3 and per_thread < (4 << 20):          # the pilot read low (start-up costs): once more, sized by this run
        per_thread = int(min(per_thread * seconds / dt, 4 << 20))
        dt = run(per_thread)
    return {"value": cores * per_thread / dt, "unit": "blocks/s", "cores": cores, "host_cores": host_cores,
            "cgroup_cpu_quota_cores": quota, "kind": "port",
            "sample": f"{per_thread} {workload} {L} B blocks per thread x {cores} threads "
                      f"(oracle/mpc_oracle.c, {'probe config' if algo == 'VPC' else algo}), {dt:.1f} s",
            "single_core_blocks_per_s": rate1}


class PowerProbe:
    """Board power and shader clock while a kernel runs back to back (OUTSIDE the timed region): the device's hwmon files
    (power1_input in microwatts, freq1_input = sclk in Hz, power1_cap) found through its PCI address, read every 10 ms
    from a thread.  DESIGN.md 4.1: the BASELINE workloads run at the board's power cap, which is what holds their clock --
    and with it the achieved fraction of the HBM peak -- down.  None when the files are not there."""

    def __init__(self, torch, device_index: int):
        import glob
        self.dir = None
        try:
            p = torch.cuda.get_device_properties(device_index)
            bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            hits = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if hits and os.path.exists(os.path.join(hits[0], "power1_input")):
                self.dir = hits[0]
        except Exception:       # noqa: BLE001 -- a reporting extra: never fails the bench
            self.dir = None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def run(self, torch, launch, stream, seconds: float = 1.0):
        if self.dir is None:
            return None
        import threading
        samples, stop = [], [False]

        def sampler():
            while not stop[0]:
                samples.append((self._read("power1_input"), self._read("freq1_input")))
                time.sleep(0.01)
        th = threading.Thread(target=sampler)
        t0 = time.perf_counter()
        launches = 0
        for _ in range(4):
            launch()
        stream.synchronize()
        th.start()
        while time.perf_counter() - t0 < seconds:
            for _ in range(8):
                launch()
                launches += 1
            stream.synchronize()
        stop[0] = True
        th.join()
        pw = [a / 1e6 for a, _ in samples if a]
        ck = [b / 1e6 for _, b in samples if b]
        cap = self._read("power1_cap")
        if not pw:
            return None
        return {"board_w_mean": sum(pw) / len(pw), "board_w_max": max(pw), "cap_w": cap / 1e6 if cap else None,
                "sclk_mhz_mean": sum(ck) / len(ck) if ck else None, "samples": len(pw), "seconds": seconds, "launches": launches,
                "source": "hwmon power1_input / freq1_input of the device, kernel launched back to back after the timed region"}


def kernel_label(mpc, ev, algo: str, L: int) -> str:
    if algo == "BDI":
        return f"bdi_kernel<{L // 4}>"
    if algo == "FPC":
        return f"fpc_kernel<{L // 4}>"
    if algo == "BPC":
        return f"bpc_kernel<{L // 4}>"
    if ev.kernel_path == mpc.MPC_PATH_VPC_FAST:
        return f"vpc_lane_kernel<{L // 4}>"        # one lane per line, W = L/4 words
    return "vpc_generic_kernel"


def make_evaluator(mpc, configs, algo: str, L: int, device: int):
    if algo == "VPC":
        return mpc.VPC(configs.probe_config(L), device=device)
    return {"BDI": mpc.BDI, "FPC": mpc.FPC, "BPC": mpc.BPC}[algo](L, device=device)


def time_workload(torch, mpc, configs, buf, stream, device, workload, algo, bytes_total, first_line=0, steps=8, power=None):
    """One sub-record: `workload` generated into (a prefix of) buf, `steps` launches timed with HIP events on
    the launch stream, the evaluator's own statistics checked for the line count."""
    kind, L = WORKLOADS[workload]
    n = bytes_total // L
    mpc.synth_fill(buf.data_ptr(), n, L, kind, first_line=first_line)
    torch.cuda.synchronize()
    ev = make_evaluator(mpc, configs, algo, L, device)
    sp = stream.cuda_stream
    for _ in range(6):                                     # warm-up
        ev.compress_device(buf.data_ptr(), n, stream=sp)
    torch.cuda.synchronize()
    ev.reset()
    ms = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        ev.compress_device(buf.data_ptr(), n, stream=sp)
        b.record(stream)
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    v = ev.stats_vector()
    assert int(v[0]) == n * steps, (workload, algo, int(v[0]), n * steps)
    check_known_ratio(algo, workload, L, n, float(v[1]) / float(v[2]))
    avg = sum(ms) / len(ms)
    achieved = n * L / (avg / 1e3) / 1e9
    rec = {"workload": workload, "algorithm": algo, "line_size": L, "blocks": n, "kernel": kernel_label(mpc, ev, algo, L),
           "kernel_ms_avg": avg, "kernel_ms_min": min(ms), "blocks_per_s": n / (avg / 1e3),
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": n * L},
           "compression_ratio": float(v[1]) / float(v[2])}
    if power is not None:
        rec["power"] = power.run(torch, lambda: ev.compress_device(buf.data_ptr(), n, stream=sp), stream, 0.8)
    ev.close()
    return rec


def layout_configs(configs, L=64):
    """VPC configurations beyond the probe's layout: non-zero RootIndex, a scan table that stops
    after 6 of the 8 bit planes, a module sequence without a built-in kernel (compiled when the handle is created)."""
    az, aws = {"name": "AllZero"}, {"name": "AllWordSame"}
    prev1 = [max(i - 1, 0) for i in range(L)]
    prev4 = [max(i - 4, 0) for i in range(L)]
    w2 = [[1.0, 0.5][i % 2] for i in range(L)]
    d1 = [1 if i % 4 == 0 else 0 for i in range(L)]
    t6 = {"TableSize": 6 * L, "Rows": [i // L for i in range(6 * L)], "Cols": [i % L for i in range(6 * L)]}
    return [
        ("probe modules, RootIndex 5/0/3/2", configs.make_config(L, [az, aws, configs.one_base(L, 5, True), configs.consecutive_base(L, 0, True),
                                                                     configs.diff_base(L, prev4, d1, 3, False), configs.weight_base(L, prev4, w2, 2, True)])),
        ("probe modules, TableSize 6 L", configs.make_config(L, [az, aws, configs.one_base(L, 0, True, t6), configs.consecutive_base(L, 0, True, t6),
                                                                 configs.diff_base(L, prev4, d1, 0, False, t6), configs.weight_base(L, prev4, w2, 0, True, t6)])),
        ("OneBase, DiffBase(i-1), WeightBase(i-4), OneBase: no built-in kernel", configs.make_config(L, [
            az, aws, configs.one_base(L, 0, True), configs.diff_base(L, prev1, [(-2 + (i % 5)) for i in range(L)], 0, False),
            configs.weight_base(L, prev4, w2, 0, True), configs.one_base(L, 0, False)])),
    ]


def time_layout(torch, mpc, buf, stream, device, name, cfg, n, L, steps=6):
    """One `layouts` record: random u32 lines (already in buf) through a VPC configuration of another layout."""
    ev = mpc.VPC(cfg, device=device)
    sp = stream.cuda_stream
    for _ in range(4):
        ev.compress_device(buf.data_ptr(), n, stream=sp)
    torch.cuda.synchronize()
    ev.reset()
    ms = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        ev.compress_device(buf.data_ptr(), n, stream=sp)
        b.record(stream)
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    v = ev.stats_vector()
    assert int(v[0]) == n * steps, (name, int(v[0]), n * steps)
    avg = sum(ms) / len(ms)
    rec = {"config": name, "workload": "random_u32", "line_size": L, "blocks": n, "kernel_form": ev.kernel_form,
           "kernel_ms_avg": avg, "kernel_ms_min": min(ms), "frac_of_hbm_peak": n * L / (avg / 1e3) / 1e9 / HBM_PEAK_GBPS,
           "compression_ratio": float(v[1]) / float(v[2])}
    ev.close()
    return rec


SUB_WORKLOADS = [("sine_f32", "VPC"), ("mixed", "VPC"), ("zeros", "VPC"), ("pointers_u64_128", "VPC"),
                 ("random_u32_32", "VPC"), ("mixed_32", "VPC"),
                 ("random_u32", "BDI"), ("sine_f32", "BDI"), ("mixed", "BDI"), ("pointers_u64_128", "BDI"), ("random_u32_32", "BDI"),
                 ("random_u32", "FPC"), ("random_u32", "BPC")]


# Known answers of the reference for the synthetic traces under the probe configuration (SURVEY.md 8c / BASELINE.md 2): the
# compression ratio is a property of the trace's period, not of its length, so it is checked at the full 16 GiB too.
KNOWN_RATIO = {
    ("VPC", "zeros", 64): 512.0 / 3.0,                        # every line 3 bits
    ("VPC", "random_u32", 64): None,                          # 512 / 515 but for a handful of lines that compress: bounded below
    ("VPC", "sine_f32", 64): 4096.0 * 512.0 / 2105344.0,      # one 4096-line period: 3456 lines at 515 b, 640 in cluster 5
    ("BDI", "pointers_u64_128", 128): 1024.0 / 564.0,         # every line B8D4
}


def check_known_ratio(algo, workload, L, n, ratio):
    key = (algo, workload, L)
    if key not in KNOWN_RATIO:
        return
    want = KNOWN_RATIO[key]
    if want is None:
        assert 512.0 / 515.0 <= ratio < 512.0 / 515.0 * (1 + 1e-6), (key, ratio)
    elif workload == "sine_f32":
        assert n % 4096 == 0 and abs(ratio - want) < 1e-12, (key, ratio, want)
    else:
        assert abs(ratio - want) < 1e-9 * want, (key, ratio, want)


def traffic_record(tj, algo, workload, L, n):
    """HBM bytes per launch from the separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes) recorded in profiles/traffic.json, with the file they were condensed from: a number
    of ANOTHER run of the same workload and build generation, not of this one."""
    e = tj.get(f"{algo}/{workload}/{L}/{n}")
    if not e:
        return None, None
    return e.get("bytes"), e.get("source")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="random_u32", choices=sorted(WORKLOADS))
    ap.add_argument("--lines", type=int, default=256 << 20, help="blocks per GPU (default 256 Mi = 16 GiB at 64 B)")
    ap.add_argument("--algo", default="VPC", choices=["VPC", "BDI", "FPC", "BPC"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true",
                    help="skip the sub-records of the other BASELINE workloads (sine, mixed, zeros, 128 B pointers; VPC and BDI)")
    ap.add_argument("--config4-lines", type=int, default=None,
                    help="N > 1: blocks per GPU of the BASELINE config 4 sub-record (mixed int/fp, 4 Gi blocks over 8 GPUs; "
                         "default 512 Mi, or --lines in a single-GPU rehearsal)")
    ap.add_argument("--rehearse-single-gpu", action="store_true",
                    help="development: run the N>1 code path with every rank on cuda:0 and the gloo backend "
                         "(a 1-GPU box cannot host an RCCL group); never used for reported numbers")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/), copied into roofline.traffic")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback path exists)")
    if args.rehearse_single_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_single_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)   # nccl == RCCL on ROCm

    mpc = importlib.import_module(PKG)
    configs = importlib.import_module(f"{PKG}.configs")
    traces = importlib.import_module(f"{PKG}.traces")
    sharded = importlib.import_module(f"{PKG}.sharded")

    kind, L = WORKLOADS[args.workload]
    n = args.lines
    buf = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mpc.synth_fill(buf.data_ptr(), n, L, kind, first_line=rank * n)
    torch.cuda.synchronize()
    ev = make_evaluator(mpc, configs, args.algo, L, local_rank)
    # a real (non-default) stream: the kernel is launched on it through the C ABI and
    # the HIP events that time it are recorded on the same stream
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sp = stream.cuda_stream

    # N > 1: the path's only exchange, a SUM all-reduce of the integer statistics, done on the device
    # accumulators (kernel -> device copy -> RCCL all-reduce on one stream, no host round trip)
    scratch = torch.zeros(ev.stats_raw_len(), dtype=torch.int64, device=dev) if world > 1 else None

    def step(ev_pair=None):
        if ev_pair is not None:
            ev_pair[0].record(stream)
        ev.compress_device(buf.data_ptr(), n, stream=sp)
        if ev_pair is not None:
            ev_pair[1].record(stream)
        if world > 1:
            sharded.all_reduce_raw_on_device(ev, scratch, sp)

    # measured streaming-read ceiling on the same buffer (first: its passes also bring the device to its working
    # clocks before the warm-up steps; two warm-up steps alone measured 6 % slower than twenty)
    probe_ms = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        mpc.read_bandwidth_probe(buf.data_ptr(), n * L, stream=sp)
        b.record(stream)
        torch.cuda.synchronize()
        probe_ms.append(a.elapsed_time(b))
    probe_gbps = n * L / (min(probe_ms[1:]) / 1e3) / 1e9

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev.reset()
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(pairs[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_single_gpu else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    kern_ms = [a.elapsed_time(b) for a, b in pairs]
    avg_kern_s = (sum(kern_ms) / len(kern_ms)) / 1e3
    # sanity: every block of every step was counted exactly once
    v = ev.stats_vector()
    assert int(v[0]) == n * args.steps, (int(v[0]), n * args.steps)
    if world > 1:
        # the all-reduced accumulators of the last step count every rank's blocks of every step
        last = ev.stats_from_raw(scratch.cpu().numpy().view(np.uint64))
        assert int(last[0]) == world * n * args.steps, (int(last[0]), world * n * args.steps)
    ratio = float(v[1]) / float(v[2])
    if rank == 0 and args.lines % 4096 == 0:
        check_known_ratio(args.algo, args.workload, L, n, ratio)

    if args.algo == "VPC" and rank == 0:
        # a full-size property no smaller run can stand in for: the byte sum behind cluster -1's MAE (VPC.cpp:417-443 with r = line[i])
        # against a sum of the resident buffer computed by torch -- they differ by at most 255 per byte of the lines that went to
        # another cluster
        total = int(buf.sum(dtype=torch.int64).item())
        count_unc, sum_r_unc = int(v[3]) // args.steps, int(v[3 + 4]) // args.steps
        assert int(v[3]) % args.steps == 0 and int(v[3 + 4]) % args.steps == 0, "every pass counts the same"
        assert abs(total - sum_r_unc) <= (n - count_unc) * L * 255, (total, sum_r_unc, n - count_unc)
    kernel_name = kernel_label(mpc, ev, args.algo, L)
    power = PowerProbe(torch, local_rank) if (world == 1 and not args.no_workloads) else None      # (not under tools/profile_round.sh)
    headline_power = power.run(torch, lambda: ev.compress_device(buf.data_ptr(), n, stream=sp), stream, 1.5) if power else None

    # ---- sub-records (not part of the timed region above) ----
    # N = 1: every other BASELINE workload through the same entry point, VPC and BDI, 16 GiB each.
    workloads = None
    if world == 1 and not args.no_workloads:
        workloads = []
        for wl, algo in SUB_WORKLOADS:
            if (wl, algo) == (args.workload, args.algo):
                continue
            workloads.append(time_workload(torch, mpc, configs, buf, stream, local_rank, wl, algo, n * L, power=power))
    # ... and the random-u32 trace through configurations of other layouts (general-layout twins, run-time compiled sequence)
    layouts = None
    if world == 1 and not args.no_workloads and args.workload == "random_u32" and L == 64:
        os.environ.setdefault("MPC_JIT_CACHE", "")      # (a measurement run leaves no code-object cache behind; ~3 s of compilation)
        mpc.synth_fill(buf.data_ptr(), n, L, kind, first_line=0)
        torch.cuda.synchronize()
        layouts = [time_layout(torch, mpc, buf, stream, local_rank, name, cfg, n, L) for name, cfg in layout_configs(configs, L)]
        # the paper figure's five data-type models (Bool/INT8, INT16, INT32/64, FP32, FP64: configs.mpc_config) at the figure's
        # 32-byte block size, at 64 and at 128 bytes (there its histogram leaves no room for the built-in kernel's rings: compiled at
        # creation for a smaller workgroup), on the same random words
        layouts.append(time_layout(torch, mpc, buf, stream, local_rank, "the paper figure's five models (configs.mpc_config), 32 B blocks",
                                   configs.mpc_config(32), n * L // 32, 32))
        layouts.append(time_layout(torch, mpc, buf, stream, local_rank, "the paper figure's five models (configs.mpc_config), 64 B blocks",
                                   configs.mpc_config(64), n, 64))
        layouts.append(time_layout(torch, mpc, buf, stream, local_rank, "the paper figure's five models (configs.mpc_config), 128 B blocks",
                                   configs.mpc_config(128), n * L // 128, 128))
    # N > 1: BASELINE config 4 -- mixed int/fp blocks sharded contiguously over the ranks, one RCCL
    # all-reduce of the statistics per pass -- as its own sub-record next to the primary value (which
    # stays on the N = 1 workload so that the scaling curve is comparable).
    config4 = None
    if world > 1:
        n4 = args.config4_lines if args.config4_lines else (min(512 << 20, args.lines) if args.rehearse_single_gpu else 512 << 20)
        del buf
        torch.cuda.empty_cache()
        buf4 = torch.empty(n4 * 64, dtype=torch.uint8, device=dev)
        mpc.synth_fill(buf4.data_ptr(), n4, 64, "mixed", first_line=rank * n4)
        torch.cuda.synchronize()
        ev4 = make_evaluator(mpc, configs, "VPC", 64, local_rank)
        scratch4 = torch.zeros(ev4.stats_raw_len(), dtype=torch.int64, device=dev)
        ks, ke, se = [], [], []
        for i in range(1 + 5):
            a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            a.record(stream)
            ev4.compress_device(buf4.data_ptr(), n4, stream=sp)
            b.record(stream)
            sharded.all_reduce_raw_on_device(ev4, scratch4, sp)
            c.record(stream)
            torch.cuda.synchronize()
            if i:
                ks.append(a.elapsed_time(b))
                se.append(a.elapsed_time(c))
        last4 = ev4.stats_from_raw(scratch4.cpu().numpy().view(np.uint64))
        assert int(last4[0]) == world * n4 * 6, (int(last4[0]), world * n4 * 6)
        t4 = torch.tensor([sum(se) / len(se), sum(ks) / len(ks)], dtype=torch.float64,
                          device="cpu" if args.rehearse_single_gpu else dev)
        dist.all_reduce(t4, op=dist.ReduceOp.MAX)
        step_ms, kern_ms4 = float(t4[0]), float(t4[1])
        config4 = {"workload": f"{world * n4} mixed int/fp 64 B blocks, {n4} per GPU, VPC probe config",
                   "sharding": f"contiguous x{world}", "rccl_ranks": dist.get_world_size(),
                   "collective": "one SUM all-reduce of the uint64 statistics accumulators per pass"
                                 + (" (gloo rehearsal)" if args.rehearse_single_gpu else " (RCCL)"),
                   "ms_per_pass": step_ms, "kernel_ms_avg": kern_ms4,
                   "all_reduce_share_of_pass": max(0.0, 1.0 - kern_ms4 / step_ms),
                   "blocks_per_s": world * n4 / (step_ms / 1e3),
                   "roofline_frac_per_gpu": n4 * 64 / (kern_ms4 / 1e3) / 1e9 / HBM_PEAK_GBPS,
                   "compression_ratio": float(last4[1]) / float(last4[2])}
        ev4.close()

    if rank == 0:
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tj = json.load(f)
        except OSError:
            tj = {}
        traffic, traffic_source = args.traffic_bytes, "--traffic-bytes"
        if traffic is None:
            traffic, traffic_source = traffic_record(tj, args.algo, args.workload, L, n)
        achieved = n * L / avg_kern_s / 1e9     # algorithmic bytes: L read per block
        out = {
            "metric": "64B blocks/s (whole node) + achieved HBM GB/s fraction; ratio bit-exact vs CPU",
            "value": world * n * args.steps / dt,
            "unit": "blocks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"{n} {args.workload} {L} B blocks per GPU, {args.algo}"
                                   + (" probe config (6 modules, all predictors)" if args.algo == "VPC" else ""),
                       "algorithm": args.algo, "line_size": L, "blocks_per_gpu": n,
                       "sharding": f"contiguous x{world}", "compression_ratio": ratio},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name,
                         "kernel_ms_avg": avg_kern_s * 1e3, "kernel_ms_min": min(kern_ms),
                         "algorithmic_bytes_per_launch": n * L,
                         "read_probe_gbps": probe_gbps, "frac_of_read_probe": achieved / probe_gbps},
        }
        if headline_power is not None:
            out["power"] = headline_power
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(configs, traces, args.workload, L, algo=args.algo)
        else:
            out["cpu_baseline"] = None
        if workloads is not None:
            for w in workloads:
                w["roofline"]["traffic"], w["roofline"]["traffic_source"] = traffic_record(
                    tj, w["algorithm"], w["workload"], w["line_size"], w["blocks"])
            out["workloads"] = workloads
        if layouts is not None:
            out["layouts"] = layouts
        if world > 1:
            # what the collective of the timed steps ran on, at the top level of the line
            out["rccl_ranks"] = dist.get_world_size()
            out["collective_backend"] = dist.get_backend() + (" (single-GPU rehearsal)" if args.rehearse_single_gpu else " (RCCL)")
        if config4 is not None:
            out["config4"] = config4
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
