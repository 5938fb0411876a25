#!/bin/bash
# Development only: build timing-ablation variants of libmpc_hip.so (results are WRONG) into
# tools/ablate/ and, on a GPU box, time each with bench.py.   tools/ablate.sh build | run
set -e
cd "$(dirname "$0")/.."
C=cal_22-mpc_amd/csrc
mkdir -p tools/ablate
if [ "$1" == "build" ]; then
  for a in 0 1 2 4 8 15; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMPC_ABLATE=$a -shared -o tools/ablate/libmpc_hip_$a.so $C/mpc_vpc_fast.hip $C/mpc_kernels.hip $C/mpc_capi.hip &
  done
  wait
else
  for a in 0 1 2 4 8 15; do
    MPC_HIP_LIB=$PWD/tools/ablate/libmpc_hip_$a.so python - <<PY
import subprocess, json, os, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--workload", os.environ.get("W", "random_u32")], capture_output=True, text=True)
line = [l for l in out.stdout.split("\n") if l.startswith("{")]
if line:
    d = json.loads(line[-1]); print("ablate $a", d["roofline"]["kernel_ms_avg"])
else:
    print("ablate $a failed", out.stdout[-300:], out.stderr[-300:])
PY
  done
fi
