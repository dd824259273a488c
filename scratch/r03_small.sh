#!/bin/bash
# Round 3: the small-launch shape (64 images of 256x256, one rank of configs[3]): conv GEMM forms, coder batches in flight; and
# the host CPU of a rank by thread.
OUT=gpurun_out/${1:-r03_j}; mkdir -p $OUT
timeout 600 python scratch/gemm_forms.py 64 256 256 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_forms_64x256.txt
timeout 300 python scratch/r03_host_cpu.py 1000 2>&1 | grep -v amdgpu.ids | tee $OUT/host_cpu.txt
for n in 3 4 5; do
  timeout 300 python bench.py --height 256 --width 256 --batch 64 --steps 60 --warmup 10 --coder-streams $n --no-cpu-baseline --no-side 2>/dev/null \
    | python scratch/r03_line.py "64x256x256 n=$n" | cut -c1-150 | tee -a $OUT/small_depth.txt
done
