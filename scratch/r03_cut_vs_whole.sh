#!/bin/bash
# product mode (three transform streams, graphs): conv GEMM launches cut at their tail (default) against whole tiles only
OUT=gpurun_out/${1:-r03_aa}; mkdir -p $OUT
for pass in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-side 2>/dev/null | python scratch/r03_line.py "default (cut where the shape calls for it)" | cut -c1-120 | tee -a $OUT/cut_vs_whole.txt
  EAE_HIP_GEMM=u timeout 300 python bench.py --no-cpu-baseline --no-side 2>/dev/null | python scratch/r03_line.py "EAE_HIP_GEMM=u (whole tiles)" | cut -c1-120 | tee -a $OUT/cut_vs_whole.txt
done
