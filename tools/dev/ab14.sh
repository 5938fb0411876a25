ulimit -c 0; mkdir -p gpurun_out
( timeout -k 10 700 python tools/ab.py run --algo BDI --rounds 2 --workloads random_u32,sine_f32,mixed,zeros,pointers_u64_128 bdis1 bdis1g16 bdis2g8 bdis1g8 bdir base ) > gpurun_out/r3_ab14.txt 2>&1
grep "FAIL" gpurun_out/r3_ab14.txt; tail -7 gpurun_out/r3_ab14.txt
