#!/bin/bash
# SQ-level picture of the headline kernel (two separate --pmc passes; no trace domains).
#   gpurun --timeout 600 -- 'bash tools/pmc_headline.sh <tag>'
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_${1:-x}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d $O/a -o a -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
  --output-format csv -d $O/b -o b -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/b.log 2>&1
python3 $R/tools/pmc_summary.py $O
