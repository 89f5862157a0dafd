#!/bin/bash
# Round evidence, every rocprofv3 pass with the PROGRAM directly after `--` (no env / bash -c hop):
#   bench line; kernel-trace stats of the bench command; separate FETCH_SIZE / WRITE_SIZE passes of
#   the bench command; per-operator kernel-trace + FETCH/WRITE passes (tools/run_op.py);
#   all configs (tools/bench_configs.py).   gpurun --timeout 1100 -- 'bash tools/collect_evidence.sh'
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence
rm -rf $O && mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --configs none --no-power > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 5 --warmup 1 --ramp-seconds 0.2 --no-cpu-baseline --configs none --no-power > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 5 --warmup 1 --ramp-seconds 0.2 --no-cpu-baseline --configs none --no-power > $O/pmc_write.log 2>&1
# per operator: the kernel-trace pass after a 1 s ramp (its averages are warm-clock numbers, thousands of calls);
# the counter passes need no ramp (bytes per launch do not depend on the clock)
for op in stft istft stftd istftd whisper gl mfcc resample mel1024 mel512 stft512; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/op_$op/kt -o kt -- python3 $R/tools/run_op.py $op 50 1.0 > $O/op_$op.kt.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/op_$op/f -o f -- python3 $R/tools/run_op.py $op 3 > $O/op_$op.f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/op_$op/w -o w -- python3 $R/tools/run_op.py $op 3 > $O/op_$op.w.log 2>&1
done
echo done
