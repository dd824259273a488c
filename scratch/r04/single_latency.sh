#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04/${1:-s51}; mkdir -p $OUT
timeout -k 10 200 python3 scratch/r04/single_latency.py < /dev/null 2>&1 | grep -v amdgpu.ids | tee $OUT/single_latency.log
cd /tmp && export TMPDIR=/tmp
N_IMAGES=30 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/scratch/r04/single_latency.py < /dev/null > $OUT/traced.log 2> $OUT/err.txt
cd $ROOT
python3 scratch/r04/single_latency.py analyse $OUT/trace | tee -a $OUT/single_latency.log
find $OUT/trace -name "*kernel_trace.csv" -delete
