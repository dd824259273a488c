#!/bin/bash
# coder batches in flight x entropy, product mode with THREE transform streams (the default since)
OUT=gpurun_out/${1:-r03_ac}; mkdir -p $OUT
for bw in 1.0 0.125 0.05 0.0125; do
  for n in 3 4 5 6 7; do
    timeout 300 python bench.py --steps 60 --warmup 8 --bin-width $bw --coder-streams $n --no-cpu-baseline --no-side 2>/dev/null \
      | python scratch/r03_line.py "bw=$bw n=$n" | cut -c1-100 | tee -a $OUT/depth3.txt
  done
done
