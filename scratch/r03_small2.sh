#!/bin/bash
# 64 images of 256x256 per step, three transform streams: coder batches in flight, graphs or launches
OUT=gpurun_out/${1:-r03_ab}; mkdir -p $OUT
for pass in 1 2; do
for n in 3 4 5 6; do
  timeout 300 python bench.py --height 256 --width 256 --batch 64 --steps 100 --warmup 10 --coder-streams $n --no-cpu-baseline --no-side 2>/dev/null \
    | python scratch/r03_line.py "64x256x256 t=3 n=$n" | cut -c1-100 | tee -a $OUT/small2.txt
done
done
timeout 300 python bench.py --height 256 --width 256 --batch 64 --steps 100 --warmup 10 --no-graphs --no-cpu-baseline --no-side 2>/dev/null \
    | python scratch/r03_line.py "64x256x256 t=3 n=auto launches" | cut -c1-100 | tee -a $OUT/small2.txt
