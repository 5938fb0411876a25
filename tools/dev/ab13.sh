ulimit -c 0; mkdir -p gpurun_out
( timeout -k 10 600 python tools/ab.py run --algo BDI --rounds 2 --workloads random_u32,sine_f32,mixed,zeros,pointers_u64_128 bdir base ) > gpurun_out/r3_ab13.txt 2>&1
grep "FAIL" gpurun_out/r3_ab13.txt; tail -3 gpurun_out/r3_ab13.txt
