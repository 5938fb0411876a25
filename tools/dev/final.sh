ulimit -c 0; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tf.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tf.txt; tail -3 gpurun_out/r3_tf.txt
rm -rf gpurun_out/prof_r03b
bash tools/profile_round.sh r03b "VPC:random_u32 VPC:sine_f32 VPC:mixed VPC:zeros VPC:pointers_u64_128 VPC:random_u32_32 VPC:mixed_32" > gpurun_out/prof_r03b.log 2>&1; tail -7 gpurun_out/prof_r03b.log
timeout -k 10 600 python bench.py > gpurun_out/r3_bench3.json 2> gpurun_out/r3_bench3.err; echo "bench rc=$?"
