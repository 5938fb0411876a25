ulimit -c 0; mkdir -p gpurun_out
( for v in w8x16 w8x10 w8x16 w8x10; do for w in random_u32 mixed zeros; do MPC_HIP_LIB=$PWD/tools/ablate/libmpc_hip_$v.so timeout -k 10 120 python tools/time_vpc.py 32 $w; done; done
  echo "== BDI (product lib)"; for w in random_u32 sine_f32 mixed; do timeout -k 10 120 python tools/time_vpc.py 64 $w BDI; done
  timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "bdi or vpc_known or test_vpc_parity or paired or deferred" 2>&1 | tail -3 ) > gpurun_out/r3_ab16.txt 2>&1
cat gpurun_out/r3_ab16.txt
