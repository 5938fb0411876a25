ulimit -c 0; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "rehearsal or gpu_shard or sharded" > gpurun_out/r3_t7.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t7.txt; tail -3 gpurun_out/r3_t7.txt
( echo "== python processes right after"; ps -eo pid,ppid,stat,etime,args --sort=pid | grep -i "python" | grep -v "grep\|GRAFT_CMD\|graft-proclimit" | cut -c1-300 ) > gpurun_out/r3_left3.txt 2>&1
sleep 5
( echo "== 5 s later"; ps -eo pid,ppid,stat,etime,args --sort=pid | grep -i "python" | grep -v "grep\|GRAFT_CMD\|graft-proclimit" | cut -c1-300 ) >> gpurun_out/r3_left3.txt 2>&1
cat gpurun_out/r3_left3.txt
timeout -k 10 600 python bench.py > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err; echo "bench rc=$?"
