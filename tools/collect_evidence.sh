#!/bin/bash
# Round evidence: GPU tests, bench line, kernel-trace stats, separate PMC passes, all configs.
# Run on the GPU box:  gpurun --timeout 1100 -- 'bash tools/collect_evidence.sh'
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -1 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json
python tools/bench_configs.py > $O/configs.json 2> $O/configs.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
echo done
