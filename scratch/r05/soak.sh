#!/bin/bash
# r05: soak of the final tree (the coder cores changed: unmasked rounds): coder fuzz against the host library next to conv GEMM launches,
# coder fuzz alone, container round trips, a short transform-chain fuzz.
mkdir -p gpurun_out/r05
L=gpurun_out/r05/${1:-soak}.log; : > $L
echo "== coder fuzz next to conv GEMM launches, 200 s" >> $L
EAE_FUZZ_LOAD=1 timeout -k 10 290 python scratch/coder_fuzz_long.py ${SEED:-51} 200 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -4 >> $L
echo "== coder fuzz alone, 100 s" >> $L
timeout -k 10 190 python scratch/coder_fuzz_long.py $((${SEED:-51}+1)) 100 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
echo "== transform chain fuzz, 90 s" >> $L
timeout -k 10 200 python scratch/transform_fuzz_long.py $((${SEED:-51}+2)) 90 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
echo "== container fuzz, 60 s" >> $L
timeout -k 10 150 python scratch/container_fuzz_long.py $((${SEED:-51}+3)) 60 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
cat $L
