ulimit -c 0; mkdir -p gpurun_out
( timeout -k 10 600 python tools/ab.py run --rounds 2 --workloads random_u32,sine_f32,mixed,zeros n16 n12 r1 base ) > gpurun_out/r3_ab9.txt 2>&1
grep "round\|==\|FAIL" gpurun_out/r3_ab9.txt; tail -5 gpurun_out/r3_ab9.txt
