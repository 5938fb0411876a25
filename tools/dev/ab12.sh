ulimit -c 0; mkdir -p gpurun_out
( timeout -k 10 600 python tools/ab.py run --rounds 2 --workloads random_u32,sine_f32,mixed,zeros n16d r1 base ) > gpurun_out/r3_ab12.txt 2>&1
grep "FAIL" gpurun_out/r3_ab12.txt; tail -4 gpurun_out/r3_ab12.txt
bash tools/dev/clk.sh "n16d:random_u32" 2>&1 | grep "==\|GRBM\|INSTS_VALU\|duration"
