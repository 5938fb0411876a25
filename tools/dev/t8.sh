ulimit -c 0; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t8.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t8.txt; tail -5 gpurun_out/r3_t8.txt
( ps -eo pid,ppid,stat,etime,args --sort=pid | grep -i "python" | grep -v "grep\|GRAFT_CMD\|graft-proclimit" | cut -c1-200; ls /dev/shm | head -3 ) > gpurun_out/r3_left4.txt 2>&1; cat gpurun_out/r3_left4.txt
timeout -k 10 300 python __graft_entry__.py --smoke 2>&1 | tail -2
