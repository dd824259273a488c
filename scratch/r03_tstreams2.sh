#!/bin/bash
# transform streams 2..5 in the product mode, the three shapes, two passes
OUT=gpurun_out/${1:-r03_x}; mkdir -p $OUT
for pass in 1 2; do
for t in 2 3 4 5; do
  timeout 300 python bench.py --transform-streams $t --no-cpu-baseline --no-side 2>/dev/null | python scratch/r03_line.py "kodak24 t=$t" | cut -c1-100 | tee -a $OUT/tstreams2.txt
done
done
for t in 2 3 4; do
  timeout 300 python bench.py --height 256 --width 256 --batch 64 --steps 60 --warmup 10 --transform-streams $t --no-cpu-baseline --no-side 2>/dev/null | python scratch/r03_line.py "64x256x256 t=$t" | cut -c1-100 | tee -a $OUT/tstreams2.txt
  timeout 300 python bench.py --height 2048 --width 2048 --batch 2 --steps 40 --warmup 8 --transform-streams $t --no-cpu-baseline --no-side 2>/dev/null | python scratch/r03_line.py "2x2048x2048 t=$t" | cut -c1-100 | tee -a $OUT/tstreams2.txt
done
