ulimit -c 0; mkdir -p gpurun_out
( timeout -k 10 600 python tools/ab.py run --rounds 2 --workloads random_u32,sine_f32,mixed cur pp256 dm12 dm4 dm0 ) > gpurun_out/r3_ab18.txt 2>&1
grep "FAIL" gpurun_out/r3_ab18.txt; tail -6 gpurun_out/r3_ab18.txt
