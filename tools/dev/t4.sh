ulimit -c 0; mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q -k "deferred_line or paired_groups" > gpurun_out/r3_t4.txt 2>&1; tail -30 gpurun_out/r3_t4.txt
