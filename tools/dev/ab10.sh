ulimit -c 0; mkdir -p gpurun_out
( timeout -k 10 600 python tools/ab.py run --rounds 2 --workloads random_u32,sine_f32,mixed,zeros n16b n16 r1 base ) > gpurun_out/r3_ab10.txt 2>&1
grep "FAIL" gpurun_out/r3_ab10.txt; tail -5 gpurun_out/r3_ab10.txt
