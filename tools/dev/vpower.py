#!/usr/bin/env python3
"""Development: board power per vector-instruction class.  Runs tools/dev/vrate in sustained mode (one instruction back to
back on every SIMD) and samples rocm-smi meanwhile.   python tools/dev/vpower.py [waves_per_simd] op ...   (GPU box)"""
import json, os, subprocess, sys, time
here = os.path.dirname(os.path.abspath(__file__))
wps = sys.argv[1]; ops = sys.argv[2:]
def smi():
    p = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True)
    try:
        c = json.loads(p.stdout); c = c[sorted(c)[0]]
        pw = [float(v) for k, v in c.items() if "ower" in k][0]
        ck = [v for k, v in c.items() if "sclk clock speed" in k][0]
        return pw, int(ck.strip("()Mhz"))
    except Exception:
        return None
print("idle", smi(), flush=True)
for op in ops:
    op, _, lanes = op.partition("@")                  # op@N: only the first N lanes of every wave run
    pr = subprocess.Popen([os.path.join(here, "vrate"), wps, "40000", op, "2.5", lanes or "0"], stdout=subprocess.PIPE, text=True)
    op = op + ("@" + lanes if lanes else "")
    time.sleep(1.0)
    got = []
    while pr.poll() is None:
        s = smi()
        if s: got.append(s)
        time.sleep(0.03)
    out = pr.stdout.read().strip()
    got = got[:-2] if len(got) > 4 else got
    pw = sum(g[0] for g in got) / max(len(got), 1); ck = sum(g[1] for g in got) / max(len(got), 1)
    rate = float(out.split(" ns per")[0].split()[-1]) if "SUSTAIN" in out else 0
    gps = float(out.split(" G wave")[0].split()[-1]) if "SUSTAIN" in out else 0
    print(f"{op:>10}: {pw:6.0f} W  sclk {ck:5.0f} MHz  {rate:.3f} ns/instr/SIMD = {rate * ck / 1000:.2f} cycles  {gps:7.1f} G wave-instr/s  ({len(got)} samples)", flush=True)
