#!/bin/bash
# transform streams 2 / 3 / 4 in the product mode (graphs) and launched kernel by kernel
OUT=gpurun_out/${1:-r03_v}; mkdir -p $OUT
for g in "" "--no-graphs"; do
for t in 2 3 4; do
  timeout 300 python bench.py --steps 60 --warmup 10 --transform-streams $t $g --no-cpu-baseline --no-side 2>/dev/null \
    | python scratch/r03_line.py "transform streams $t $g" | cut -c1-140 | tee -a $OUT/tstreams.txt
done
done
