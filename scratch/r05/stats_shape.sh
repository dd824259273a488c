#!/bin/bash
# kernel-trace stats of one short bench run at a given shape: bash scratch/r05/stats_shape.sh <tag> <bench flags...>
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-side --no-dropin-surface --steps 30 --warmup 5 --min-seconds 0 --max-blocks 1 "$@" > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cd $ROOT
s=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$s" $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -delete
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[1:26]:
    print(r[0][:80].ljust(80), r[1].rjust(6), '%9.1f us avg' % (float(r[3])/1e3), '%6.2f%%' % float(r[4]))
PY
