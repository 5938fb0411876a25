ulimit -c 0; mkdir -p gpurun_out
( for g in 1024 2048 4096 16384; do echo "== MPC_TEST_GRID=$g (w16: $((g/4)) workgroups)"; MPC_TEST_GRID=$g timeout -k 10 200 python tools/ab.py run --rounds 1 --workloads random_u32,sine_f32,zeros w16s2t w16s1t; done ) > gpurun_out/r3_ab8.txt 2>&1
grep "round\|==\|FAIL" gpurun_out/r3_ab8.txt
