#!/bin/bash
# GPU box: rocprofv3 evidence for bench.py workloads -> gpurun_out/prof_$TAG/ (copy what is judged into profiles/ with
# tools/profile_summarize.py).
#   tools/profile_round.sh TAG "VPC:random_u32 VPC:mixed BDI:sine_f32 ..."
# Per workload 16 GiB resident (the size of bench.py's records: 512 Mi / 256 Mi / 128 Mi lines of 32 / 64 / 128 bytes), one
# kernel-trace pass and three --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ counters), each its own run (counters are never
# combined with trace domains other than --kernel-trace / --stats; the program itself follows `--`).
set -e
TAG=$1; LIST=$2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for aw in $LIST; do
  ALGO=${aw%%:*}; w=${aw##*:}
  case $w in *_128) L=128;; *_32) L=32;; *) L=64;; esac
  N=$(( (16 << 30) / L ))
  B="python3 $R/bench.py --workload $w --algo $ALGO --lines $N --no-cpu-baseline --no-workloads --steps 10 --warmup 2"
  # the kernel-trace pass runs 40 timed steps so that the two warm-up launches weigh little in the summary's average
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${ALGO}_${w}_trace -o run -- python3 $R/bench.py --workload $w --algo $ALGO --lines $N --no-cpu-baseline --no-workloads --steps 40 --warmup 2 > $OUT/${ALGO}_${w}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${ALGO}_${w}_fetch -o run -- $B > $OUT/${ALGO}_${w}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${ALGO}_${w}_write -o run -- $B > $OUT/${ALGO}_${w}_write.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${ALGO}_${w}_sq -o run -- $B > $OUT/${ALGO}_${w}_sq.log 2>&1
  echo "done $ALGO $w ($N lines): $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/${ALGO}_${w}_trace.log | head -1)"
done
