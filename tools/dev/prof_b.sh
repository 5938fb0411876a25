ulimit -c 0
bash tools/profile_round.sh r03a "BDI:random_u32 BDI:sine_f32 BDI:mixed BDI:pointers_u64_128 BDI:random_u32_32 FPC:random_u32 BPC:random_u32" > gpurun_out/prof_r03b.log 2>&1; tail -8 gpurun_out/prof_r03b.log
