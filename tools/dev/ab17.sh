ulimit -c 0; mkdir -p gpurun_out
( echo "== new (LK_ANY instantiations)"; MPC_HIP_LIB=$PWD/tools/ablate/libmpc_hip_any.so timeout -k 10 300 python tools/time_layouts.py 2>&1 | grep -v amdgpu.ids
  echo "== n16d (run-time loop for these layouts)"; MPC_HIP_LIB=$PWD/tools/ablate/libmpc_hip_n16d.so timeout -k 10 300 python tools/time_layouts.py 2>&1 | grep -v amdgpu.ids
  MPC_HIP_LIB=$PWD/tools/ablate/libmpc_hip_any.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "nonzero_root_and_truncated and 64" 2>&1 | tail -3 ) > gpurun_out/r3_ab17.txt 2>&1
cat gpurun_out/r3_ab17.txt
