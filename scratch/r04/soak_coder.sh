#!/bin/bash
# the coder legs of soak.sh again (after the coder kernels went back to their own register allocations): fuzz next to conv GEMM launches, then alone
mkdir -p gpurun_out/r04
L=gpurun_out/r04/${1:-s37}_soak_coder.log; : > $L
echo "== coder fuzz next to conv GEMM launches, 240 s" >> $L
EAE_FUZZ_LOAD=1 timeout -k 10 330 python scratch/coder_fuzz_long.py 51 240 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -4 >> $L
echo "== coder fuzz alone, 100 s" >> $L
timeout -k 10 200 python scratch/coder_fuzz_long.py 52 100 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
cat $L
