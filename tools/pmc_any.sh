#!/bin/bash
# SQ picture of one operator's kernels:  gpurun -- 'bash tools/pmc_any.sh <tag> <op> <kernel-substring> <units per launch>'
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcany_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OP=$2
run() { rocprofv3 --pmc "${@:2}" --output-format csv -d $O/$1 -o $1 -- python3 $R/tools/run_op.py $OP 4 > $O/$1.log 2>&1; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
python3 $R/tools/pmc_summary.py $O $3 $4
