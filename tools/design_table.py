#!/usr/bin/env python3
"""Development: the measurement table of DESIGN.md section 6 from profiles/rNN_bench_line.json (one bench.py line) and
the tracked rocprofv3 summaries / counter passes of the same round.   python tools/design_table.py 3"""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d = json.load(open(os.path.join(ROOT, "profiles", f"r{rnd:02d}_bench_line.json")))


def find(kind, algo, wl):
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r{rnd:02d}_*_{kind}_{algo}_{wl}.csv")))
    return hits[-1] if hits else None


def prof_ms(algo, wl):
    p = find("kernel_stats", algo, wl)
    if p:
        for row in csv.DictReader(open(p)):
            if re.search(r"(vpc_lane|bdi|fpc|bpc)_kernel", row["Name"]):
                return float(row["AverageNs"]) / 1e6


def pmc(algo, wl, name):
    p = find("pmc", algo, wl)
    if p:
        for row in csv.DictReader(open(p)):
            if row["counter"] == name:
                return float(row["value_per_launch"])


rows = [("random_u32", "VPC", d["roofline"]["kernel"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["roofline"]["traffic"],
         d["roofline"]["algorithmic_bytes_per_launch"], d["config"]["compression_ratio"], d["config"]["blocks_per_gpu"])]
for w in d["workloads"]:
    rows.append((w["workload"], w["algorithm"], w["kernel"], w["kernel_ms_avg"], w["roofline"]["frac"], w["roofline"]["traffic"],
                 w["roofline"]["algorithmic_bytes_per_launch"], w["compression_ratio"], w["blocks"]))
print("| workload | algorithm | kernel | bench ms / 16 GiB | of 8 TB/s | rocprofv3 avg of 42 (ms) | HBM traffic / algorithmic | vector instr. per 64 lines | ratio |")
print("|---|---|---|---|---|---|---|---|---|")
for wl, algo, kern, ms, fr, tr, alg, ratio, n in rows:
    valu, pm = pmc(algo, wl, "SQ_INSTS_VALU"), prof_ms(algo, wl)
    print(f"| {wl} | {algo} | `{kern}` | {ms:.3f} | **{fr:.3f}** | {pm:.3f} | {tr / alg if tr else float('nan'):.4f} | {valu / (n / 64):.0f} | {ratio:.4f} |")
