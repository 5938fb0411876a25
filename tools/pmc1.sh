#!/bin/bash
# Development helper (GPU box): one SQ counter pass.  tools/pmc1.sh OUTDIR WORKLOAD [ALGO]   (MPC_HIP_LIB honoured)
set -e
out=$1; w=$2; a=${3:-VPC}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM -d $R/$out/p1 -o run -- python3 $R/tools/time_vpc.py 64 $w $a > $R/$out/p1.log 2>&1
python3 $R/tools/pmc_sum.py $R/$out/p1 | grep "INSTS\|duration"
