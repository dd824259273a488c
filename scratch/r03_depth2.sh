#!/bin/bash
OUT=gpurun_out/r03_h; mkdir -p $OUT
for bw in 1.0 0.125 0.05 0.0125; do
  for n in 3 4 5 6; do
    timeout 300 python bench.py --steps 40 --warmup 8 --bin-width $bw --coder-streams $n --no-cpu-baseline --no-side 2>/dev/null \
      | python scratch/r03_line.py "bw=$bw n=$n" | cut -c1-120 | tee -a $OUT/depth.txt
  done
done
