#!/bin/bash
# rocprofv3 kernel-trace stats of one operator:  gpurun -- 'bash tools/kt_op.sh <tag> <op>'
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kt_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/tools/run_op.py $2 4 > $O/log.txt 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}%")
PY
