#!/bin/bash
# r04: soak of the final tree: coder fuzz against the host library next to conv GEMM launches; coder fuzz alone; the whole transform
# chain against the C oracle over random shapes / models / stream and graph settings; container round trips.
mkdir -p gpurun_out/r04
L=gpurun_out/r04/${1:-s24}_soak.log; : > $L
echo "== coder fuzz next to conv GEMM launches, 240 s" >> $L
EAE_FUZZ_LOAD=1 timeout -k 10 330 python scratch/coder_fuzz_long.py ${SEED:-41} 240 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -4 >> $L
echo "== coder fuzz alone, 150 s" >> $L
timeout -k 10 240 python scratch/coder_fuzz_long.py $((${SEED:-41}+1)) 150 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
echo "== transform chain fuzz, 300 s" >> $L
timeout -k 10 420 python scratch/transform_fuzz_long.py $((${SEED:-41}+2)) 300 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
echo "== container fuzz, 90 s" >> $L
timeout -k 10 180 python scratch/container_fuzz_long.py $((${SEED:-41}+3)) 90 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
cat $L
