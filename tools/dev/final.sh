ulimit -c 0; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tf.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tf.txt; tail -3 gpurun_out/r3_tf.txt
timeout -k 10 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/r3_bench4.json 2> gpurun_out/r3_bench4.err; echo "bench rc=$?"
