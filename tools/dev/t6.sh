ulimit -c 0; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t6.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t6.txt; tail -4 gpurun_out/r3_t6.txt
( echo "== processes after the GPU test step"; ps -eo pid,ppid,stat,etime,comm --sort=pid | tail -15; echo "== /dev/shm"; ls -la /dev/shm | head; echo "== /tmp"; ls -la /tmp | head -20 ) > gpurun_out/r3_left2.txt 2>&1
rm -rf gpurun_out/prof_r03a
bash tools/profile_round.sh r03a "VPC:random_u32 VPC:sine_f32 VPC:mixed VPC:zeros VPC:pointers_u64_128 VPC:random_u32_32 VPC:mixed_32 BDI:random_u32 BDI:sine_f32 BDI:mixed BDI:pointers_u64_128 BDI:random_u32_32 FPC:random_u32 BPC:random_u32" > gpurun_out/prof_r03a.log 2>&1; tail -15 gpurun_out/prof_r03a.log
