#!/bin/bash
# Development only: build variants of libmpc_hip.so into tools/ablate/ and, on a GPU box,
# time each with bench.py IN ONE CALL (boxes differ by several percent, so only same-call
# comparisons mean anything).
#   FLAG=MPC_ABLATE VARIANTS="0 1 2 4 8 15" tools/ablate.sh build   (timing ablations: WRONG results)
#   W=random_u32 VARIANTS="a b a b" tools/ablate.sh run
set -e
cd "$(dirname "$0")/.."
C=cal_22-mpc_amd/csrc
mkdir -p tools/ablate
V=${VARIANTS:-0 1 2 4 8 15}
if [ "$1" == "build" ]; then
  rm -f tools/ablate/*.so
  for a in $V; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D${FLAG:-MPC_ABLATE}=$a -shared -o tools/ablate/libmpc_hip_$a.so $C/mpc_vpc_lane.hip $C/mpc_kernels.hip $C/mpc_capi.hip &
  done
  wait
else
  for a in $V; do
    MPC_HIP_LIB=$PWD/tools/ablate/libmpc_hip_$a.so python - <<PY
import subprocess, json, os, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--workload", os.environ.get("W", "random_u32")], capture_output=True, text=True)
line = [l for l in out.stdout.split("\n") if l.startswith("{")]
if line:
    d = json.loads(line[-1]); print("variant $a", d["roofline"]["kernel_ms_avg"])
else:
    print("variant $a failed", out.stdout[-300:], out.stderr[-300:])
PY
  done
fi
