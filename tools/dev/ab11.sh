ulimit -c 0; mkdir -p gpurun_out
( timeout -k 10 600 python tools/ab.py run --rounds 1 --workloads random_u32,sine_f32,mixed,zeros n16c n16b r1 ) > gpurun_out/r3_ab11.txt 2>&1
grep "FAIL\|round" gpurun_out/r3_ab11.txt
bash tools/dev/clk.sh "n16c:random_u32" 2>&1 | grep "==\|GRBM\|INSTS_VALU\|duration"
