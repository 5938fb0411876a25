#!/usr/bin/env python3
"""Condense gpurun_out/prof_TAG/ (tools/profile_round.sh) into the tracked evidence files:

    profiles/rNN_TAG_kernel_stats_ALGO_WORKLOAD.csv   rocprofv3 --kernel-trace --stats summary (copied)
    profiles/rNN_TAG_pmc_ALGO_WORKLOAD.csv            FETCH_SIZE / WRITE_SIZE / SQ counters per launch, one pass each
    profiles/traffic.json                             HBM bytes per launch that bench.py copies into roofline.traffic

    python tools/profile_summarize.py TAG ROUND [lines_per_launch]
"""
import csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], int(sys.argv[2])
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
KERNELS = ("vpc_lane_kernel", "vpc_generic_kernel", "bdi_kernel", "fpc_kernel", "bpc_kernel")
LINES = {32: 512 << 20, 64: 256 << 20, 128: 128 << 20}      # 16 GiB resident per workload (tools/profile_round.sh)


def counters(path):
    acc = {}
    for r in csv.DictReader(open(path)):
        if any(k in r["Kernel_Name"] for k in KERNELS):
            acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    out = {}
    for name, per in acc.items():
        v = [per[k] for k in sorted(per, key=int)]
        v = v[2:] if len(v) > 4 else v          # drop the warm-up launches
        out[name] = (sum(v) / len(v), len(v))
    return out


tj_path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tj_path))
tj = {k: v for k, v in tj.items() if isinstance(v, dict) and v.get("round") == rnd}      # entries of earlier rounds belong to other kernels
for tr in sorted(glob.glob(os.path.join(src, "*_trace"))):
    base = os.path.basename(tr)[: -len("_trace")]          # ALGO_workload
    algo, wl = base.split("_", 1)
    L = 128 if wl.endswith("_128") else (32 if wl.endswith("_32") else 64)
    n = LINES[L]
    dst = os.path.join(ROOT, "profiles", f"r{rnd:02d}_{tag}_kernel_stats_{base}.csv")
    shutil.copy(os.path.join(tr, "run_kernel_stats.csv"), dst)
    rows = [("pass", "kernel", "counter", "value_per_launch", "launches", "note")]
    kname = None
    fetch = write = None
    for kind in ("fetch", "write", "sq"):
        p = os.path.join(src, f"{base}_{kind}", "run_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            if any(k in r["Kernel_Name"] for k in KERNELS):
                m = re.search(r"((?:vpc_lane|vpc_generic|bdi|fpc|bpc)_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
                kname = m.group(1).replace(", ", ";") if m else r["Kernel_Name"][:60]
                break
        for name, (val, cnt) in sorted(counters(p).items()):
            note = ""
            if name == "FETCH_SIZE":
                fetch = val
                note = "KiB; gfx950 reports 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md, HBM): bytes = 2*value*1024"
            elif name == "WRITE_SIZE":
                write = val
                note = "KiB"
            elif name.startswith("SQ_"):
                note = f"{val / (n / 64):.2f} per group of 64 lines"
            elif name == "GRBM_GUI_ACTIVE":
                note = "sum over the 8 XCDs: effective clock = value / 8 / kernel time"
            rows.append((f"{tag}_{kind}", kname, name, f"{val:.3f}", cnt, note))
    with open(os.path.join(ROOT, "profiles", f"r{rnd:02d}_{tag}_pmc_{base}.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows)
    if fetch is not None and write is not None:
        tj[f"{algo}/{wl}/{L}/{n}"] = {"fetch_size_kib": fetch, "write_size_kib": write,
                                      "bytes": int(round(2 * fetch * 1024 + write * 1024)), "round": rnd,
                                      "source": f"profiles/r{rnd:02d}_{tag}_pmc_{base}.csv"}
    print(base, "traffic bytes", tj.get(f"{algo}/{wl}/{L}/{n}", {}).get("bytes"), "algorithmic", n * L)
json.dump(tj, open(tj_path, "w"), indent=2)
