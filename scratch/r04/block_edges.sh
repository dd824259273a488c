#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04/${1:-s42}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 20 --warmup 5 --min-seconds 0 --max-blocks 3 --no-cpu-baseline --no-side < /dev/null > $OUT/bench.json 2> $OUT/err.txt
cd $ROOT
python3 scratch/r04/block_edges.py $OUT/trace | tee $OUT/block_edges.log
find $OUT/trace -name "*kernel_trace.csv" -delete
