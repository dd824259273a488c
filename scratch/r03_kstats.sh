#!/bin/bash
# kernel stats of the default bench command at two latent entropies (rocprofv3 --kernel-trace --stats), summaries only
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03_g; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for bw in 1.0 0.05; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$bw -- python3 $ROOT/bench.py --bin-width $bw --no-cpu-baseline --no-side > $OUT/bench_$bw.json 2> $OUT/trace_$bw.err
  s=$(find $OUT/trace_$bw -name "*kernel_stats.csv" | head -1); cp "$s" $OUT/kernel_stats_$bw.csv
  find $OUT/trace_$bw -name "*kernel_trace.csv" -delete
  echo "== bw $bw"; head -16 $OUT/kernel_stats_$bw.csv | cut -c1-150
done
