ulimit -c 0; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t5.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t5.txt
( echo "== processes after the GPU test step"; ps -eo pid,ppid,stat,etime,comm,args --sort=pid | grep -v "ps -eo" | tail -40
  echo "== /dev/shm"; ls -la /dev/shm | head -30
  echo "== /tmp"; ls -la /tmp | head -40
  echo "== gpu pids"; rocm-smi --showpids 2>&1 | tail -15 ) > gpurun_out/r3_left.txt 2>&1
tail -6 gpurun_out/r3_t5.txt
timeout -k 10 600 python bench.py > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r3_bench1.err
