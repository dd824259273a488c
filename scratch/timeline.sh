#!/bin/bash
# rocprofv3 kernel trace of a short bench run; prints one steady-state step (scratch/timeline.py). Extra bench flags: "$@"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 12 --warmup 6 --min-seconds 0 --no-cpu-baseline --no-single-image "$@" > $OUT/bench.json 2> $OUT/err.txt
cd $ROOT
python3 scratch/timeline.py $OUT/trace
find $OUT/trace -name "*kernel_trace.csv" -delete
