#!/usr/bin/env python3
"""Development: board power and shader clock while one kernel runs back to back for a few seconds.

    python tools/dev/power.py [--seconds 4] [--algo VPC] random_u32 sine_f32 mixed zeros   (on the GPU box)

Samples `rocm-smi --showpower --showclocks --json` from a thread while the evaluator's kernel is launched in a loop
on 16 GiB of resident lines; prints per workload the mean kernel time, mean / max power and the sclk readings.
"""
import importlib, json, os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mpc = importlib.import_module("cal_22-mpc_amd")
configs = importlib.import_module("cal_22-mpc_amd.configs")

args = sys.argv[1:]
seconds, algo = 4.0, "VPC"
if "--seconds" in args:
    i = args.index("--seconds"); seconds = float(args[i + 1]); del args[i:i + 2]
if "--algo" in args:
    i = args.index("--algo"); algo = args[i + 1]; del args[i:i + 2]
W = {"random_u32": ("random_u32", 64), "sine_f32": ("sine_f32", 64), "mixed": ("mixed", 64), "zeros": ("zeros", 64),
     "pointers_u64_128": ("pointers_u64", 128), "random_u32_32": ("random_u32", 32), "mixed_32": ("mixed", 32)}
samples, stop = [], False

def smi():
    p = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True)
    try:
        d = json.loads(p.stdout)
        c = d[sorted(d)[0]]
        return {k: v for k, v in c.items() if "ower" in k or "sclk" in k or "mclk" in k or "fclk" in k}
    except Exception as e:      # noqa
        return {"error": (p.stdout + p.stderr)[-300:]}

def sampler():
    while not stop:
        samples.append(smi())
        time.sleep(0.05)

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
n = 256 << 20
print("idle:", smi(), flush=True)
for w in args:
    kind, L = W[w]
    nl = n * 64 // L
    buf = torch.empty(nl * L, dtype=torch.uint8, device=dev)
    mpc.synth_fill(buf.data_ptr(), nl, L, kind, first_line=0)
    ev = mpc.VPC(configs.probe_config(L)) if algo == "VPC" else getattr(mpc, algo)(L)
    for _ in range(3):
        ev.compress_device(buf.data_ptr(), nl, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); launches = 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    while time.time() - t0 < seconds:
        for _ in range(20):
            ev.compress_device(buf.data_ptr(), nl, stream=stream.cuda_stream); launches += 1
        stream.synchronize()
    b.record(stream); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / launches
    stop = True; th.join()
    pw = [float(v) for s in samples for k, v in s.items() if "ower" in k and str(v).replace(".", "").isdigit()]
    sclk = [v for s in samples for k, v in s.items() if "sclk" in k]
    print(f"{w:>18} {algo}: {ms:.3f} ms/launch  power mean {sum(pw) / max(len(pw), 1):.0f} W max {max(pw or [0]):.0f} W  "
          f"({len(pw)} samples)  sclk seen {sorted(set(sclk))[:6]}", flush=True)
    if samples: print("    last sample:", samples[-1], flush=True)
    ev.close(); del buf
