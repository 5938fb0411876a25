ulimit -c 0
bash tools/profile_round.sh r03a "VPC:random_u32 VPC:sine_f32 VPC:mixed VPC:zeros VPC:pointers_u64_128 VPC:random_u32_32 VPC:mixed_32" > gpurun_out/prof_r03a.log 2>&1; tail -8 gpurun_out/prof_r03a.log
