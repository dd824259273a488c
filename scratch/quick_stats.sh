#!/bin/bash
# Kernel-trace stats of the default bench command only (no PMC passes): bash scratch/quick_stats.sh <tag>
TAG=${1:-quick}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-single-image > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cd $ROOT
s=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$s" $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -delete
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[1:18]:
    print(r[0][:72].ljust(72), r[1].rjust(5), '%9.1f us avg' % (float(r[3])/1e3), '%6.2f%%' % float(r[4]))
PY
