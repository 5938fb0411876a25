#!/usr/bin/env python3
"""Development: compile one translation unit for gfx950 with -Rpass-analysis=kernel-resource-usage and print
one line per kernel (VGPRs, AGPRs, spilled VGPRs/SGPRs, scratch bytes per lane, occupancy, LDS).

    tools/kres.py cal_22-mpc_amd/csrc/mpc_vpc_lane.hip -DMPC_LANE_W=16 [--filter lane_kernelILi16ELb0] [--save profiles/x.txt]
"""
import os
import re
import subprocess
import sys
import tempfile

args = sys.argv[1:]
flt = save = None
if "--filter" in args:
    i = args.index("--filter"); flt = args[i + 1]; del args[i:i + 2]
if "--save" in args:
    i = args.index("--save"); save = args[i + 1]; del args[i:i + 2]
src, extra = args[0], args[1:]
with tempfile.TemporaryDirectory() as td:
    p = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", *extra,
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(td, "o.o")],
                       capture_output=True, text=True)
if p.returncode:
    sys.stderr.write(p.stderr[-4000:])
    raise SystemExit(p.returncode)
rows, cur = [], None
for line in p.stderr.split("\n"):
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass-analysis", line) or re.search(r":\d+:\d+: remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
demangle = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n") if rows else []
out = []
for r, d in zip(rows, demangle):
    if flt and flt not in r["name"] and flt not in d:
        continue
    short = re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0].replace("void ", "")
    out.append(f"{short:<58} VGPR {r.get('VGPRs', '?'):>3} AGPR {r.get('AGPRs', '?'):>3} spillV {r.get('VGPRs Spill', '?'):>3} "
               f"spillS {r.get('SGPRs Spill', '?'):>3} scratch {r.get('ScratchSize [bytes/lane]', '?'):>4} occ {r.get('Occupancy [waves/SIMD]', '?')} "
               f"SGPR {r.get('TotalSGPRs', r.get('SGPRs', '?'))} LDS {r.get('LDS Size [bytes/block]', '?')}")
text = "\n".join(out)
print(text)
if save:
    with open(save, "w") as f:
        f.write(f"# hipcc -O3 --offload-arch=gfx950 {' '.join(extra)} -Rpass-analysis=kernel-resource-usage {src}\n{text}\n")
