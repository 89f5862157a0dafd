#!/bin/bash
# SQ + GRBM picture of the n_fft=2048 mel kernels (variant chosen by the caller's environment):
#   gpurun -- 'AP_MEL2048_WAVE=1 bash tools/pmc_mel.sh v1 mel2048'
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcmel_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --pmc "${@:2}" --output-format csv -d $O/$1 -o $1 -- python3 $R/tools/run_op.py mel 4 > $O/$1.log 2>&1; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
python3 $R/tools/pmc_summary.py $O $2 110336
