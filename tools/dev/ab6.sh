ulimit -c 0; mkdir -p gpurun_out
( timeout -k 10 300 python tools/ab.py run --rounds 1 --workloads random_u32,sine_f32,mixed,zeros r1 r1c ) > gpurun_out/r3_ab6.txt 2>&1
grep "round\|==" gpurun_out/r3_ab6.txt
