ulimit -c 0; mkdir -p gpurun_out
( timeout -k 10 500 python tools/ab.py run --rounds 2 --workloads random_u32,sine_f32,mixed,zeros r1 r1t w16s2t w16s1t w16s2 ) > gpurun_out/r3_ab7.txt 2>&1
grep "round\|==\|FAIL" gpurun_out/r3_ab7.txt; tail -6 gpurun_out/r3_ab7.txt
