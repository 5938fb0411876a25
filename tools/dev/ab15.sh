ulimit -c 0; mkdir -p gpurun_out
( for a in BDI FPC BPC; do echo "== $a"; timeout -k 10 400 python tools/ab.py run --algo $a --rounds 2 --workloads random_u32,sine_f32,mixed,zeros,pointers_u64_128 kr base $( [ $a = BDI ] && echo bdis1 ); done ) > gpurun_out/r3_ab15.txt 2>&1
grep "FAIL\|==\|: random" gpurun_out/r3_ab15.txt | grep -v round
