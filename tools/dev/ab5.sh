ulimit -c 0; mkdir -p gpurun_out
( timeout -k 10 300 python tools/ab.py run --rounds 1 --workloads random_u32,zeros r1 r1q w8s1 w8s2 w16s1
  echo "== MPC_TEST_GRID=1024 (r1: 4 WGs per CU once; w16s1: 1024 WGs)"
  MPC_TEST_GRID=1024 timeout -k 10 200 python tools/ab.py run --rounds 1 --workloads random_u32,zeros r1 w16s1
  echo "== MPC_TEST_GRID=2048"
  MPC_TEST_GRID=2048 timeout -k 10 200 python tools/ab.py run --rounds 1 --workloads random_u32,zeros r1 w16s1
  echo "== MPC_TEST_GRID=4096"
  MPC_TEST_GRID=4096 timeout -k 10 200 python tools/ab.py run --rounds 1 --workloads random_u32,zeros r1 w16s1 ) > gpurun_out/r3_ab5.txt 2>&1
grep -v "^----\|mean of" gpurun_out/r3_ab5.txt | grep "round\|==" 
