#!/bin/bash
# GPU box: rocprofv3 evidence for bench.py workloads -> gpurun_out/prof_$TAG/ (copy what is judged into profiles/).
#   tools/profile_round.sh TAG "random_u32 mixed sine_f32" [ALGO]
# One kernel-trace pass and three --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ counters) per workload, each its own run
# (counters are never combined with trace domains other than --kernel-trace / --stats).
set -e
TAG=$1; WLS=${2:-"random_u32 mixed sine_f32"}; ALGO=${3:-VPC}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in $WLS; do
  B="python3 $R/bench.py --workload $w --algo $ALGO --no-cpu-baseline --no-workloads --steps 10 --warmup 2"
  # the kernel-trace pass runs 40 timed steps so that the two warm-up launches weigh little in the summary's average
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${ALGO}_${w}_trace -o run -- python3 $R/bench.py --workload $w --algo $ALGO --no-cpu-baseline --no-workloads --steps 40 --warmup 2 > $OUT/${ALGO}_${w}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${ALGO}_${w}_fetch -o run -- $B > $OUT/${ALGO}_${w}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${ALGO}_${w}_write -o run -- $B > $OUT/${ALGO}_${w}_write.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/${ALGO}_${w}_sq -o run -- $B > $OUT/${ALGO}_${w}_sq.log 2>&1
  echo "done $ALGO $w"
done
find $OUT -name "*.csv" | head -40
