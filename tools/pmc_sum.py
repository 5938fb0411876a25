#!/usr/bin/env python3
"""Development helper: per-kernel averages of rocprofv3 --pmc results (rocpd .db or counter_collection.csv).
   python tools/pmc_sum.py DIR [kernel-substring] [lines-per-launch]"""
import csv, glob, os, sys, collections, sqlite3
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "vpc_lane_kernel"; n = int(sys.argv[3]) if len(sys.argv) > 3 else 256 << 20
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
    c = sqlite3.connect(f)
    for name, disp, val in c.execute("select counter_name, dispatch_id, sum(value) from counters_collection "
                                     "where kernel_name like ? group by counter_name, dispatch_id order by dispatch_id", (f"%{sub}%",)):
        acc[name].append(float(val))
    dur = [r[0] for r in c.execute("select duration from kernels where name like ? order by start", (f"%{sub}%",))]
    if dur:
        acc["(kernel duration, us)"] = [x / 1e3 for x in dur]
for k in sorted(acc):
    v = acc[k]; v = v[1:] if len(v) > 2 else v      # drop the first (warm-up) dispatch
    m = sum(v) / len(v)
    print(f"{k:28s} {m:16.1f} per launch   {m / (n / 64):10.2f} per group of 64 lines   ({len(v)} launches)")
