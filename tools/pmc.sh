#!/bin/bash
# Development helper (GPU box): SQ counter passes of the VPC kernel.  tools/pmc.sh OUTDIR WORKLOAD [ALGO]
# MPC_HIP_LIB may point at a variant library (export it before calling).
set -e
out=$1; w=$2; a=${3:-VPC}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $R/$out/p1 -o run -- python3 $R/tools/time_vpc.py 64 $w $a > $R/$out/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM -d $R/$out/p2 -o run -- python3 $R/tools/time_vpc.py 64 $w $a > $R/$out/p2.log 2>&1 || true
python3 $R/tools/pmc_sum.py $R/$out/p1 | tee $R/$out/summary.txt
python3 $R/tools/pmc_sum.py $R/$out/p2 | tee -a $R/$out/summary.txt
