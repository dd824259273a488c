#!/bin/bash
mkdir -p gpurun_out/r04
L=gpurun_out/r04/s5_accum.log; : > $L
for v in a_n48_acc40 a_n48_acc44 a_n44_acc44 a_n41_acc40; do
  EAE_HIP_LIB=scratch/r04/libs/$v/libeae_hip.so timeout 300 python scratch/r04/decode_hunt.py 1.0 4 "none,VALU only,MFMA only" 2>&1 | grep -v "amdgpu.ids\|LDS 163840" | cut -c1-200 >> $L
done
EAE_HIP_LIB=scratch/r04/libs/a_lds/libeae_hip.so timeout 300 python scratch/r04/decode_hunt.py 1.0 4 "none,VALU only,MFMA only,conv GEMM" 2>&1 | grep -v "amdgpu.ids" | cut -c1-200 >> $L
cat $L
