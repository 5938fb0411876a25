ulimit -c 0; mkdir -p gpurun_out
bash tools/dev/clk.sh "cur:mixed a16:mixed a8:mixed a4:mixed cur:sine_f32 a16:sine_f32 a8:sine_f32" 2>&1 | grep "==\|INSTS_VALU\|duration\|GRBM" > gpurun_out/r3_abl.txt; cat gpurun_out/r3_abl.txt
