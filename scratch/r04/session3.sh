#!/bin/bash
# r04 GPU session 3: the failing decoder core with ONLY its VGPR allocation changed in the assembly (40 -> 48 / 56 / 64), and the
# shipped core with its allocation raised from 48 to 56 / 64
mkdir -p gpurun_out/r04
L=gpurun_out/r04/s3_vgpr.log; : > $L
for v in a_identity a_vgpr48 a_vgpr56 a_vgpr64 ship0_vgpr56 ship0_vgpr64 ship3_vgpr56; do
  EAE_HIP_LIB=scratch/r04/libs/$v/libeae_hip.so timeout 300 python scratch/r04/decode_hunt.py 1.0 4 "none,conv GEMM,VALU only,MFMA only" 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $L
done
cat $L
