ulimit -c 0; mkdir -p gpurun_out
timeout -k 10 600 python tools/ab.py run --rounds 2 --workloads random_u32_32,mixed_32,random_u32 $AB_VARIANTS 2>&1 | grep "FAIL\|mean\|: random" | grep -v round
