#!/bin/bash
# Development (GPU box): effective clock and issue counters of variant builds: tools/dev/clk.sh "variant:workload ..."
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/clk; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for vw in $1; do
  v=${vw%%:*}; w=${vw##*:}
  export MPC_HIP_LIB=$R/tools/ablate/libmpc_hip_$v.so
  rm -rf $OUT/${v}_${w}
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $OUT/${v}_${w} -o run -- python3 $R/tools/time_vpc.py 64 $w > $OUT/${v}_${w}.log 2>&1
  echo "== $v $w: $(grep 'ms / 16' $OUT/${v}_${w}.log)"
  python3 $R/tools/pmc_sum.py $OUT/${v}_${w}
done
