#!/bin/bash
mkdir -p gpurun_out/r04
L=gpurun_out/r04
EAE_HUNT_PLACEMENT=1 EAE_HIP_LIB=scratch/r04/libs/topup_p0/libeae_hip.so timeout 600 python scratch/r04/decode_hunt.py 1.0 3 "none,conv GEMM,VALU only,memory copy,MFMA only, LDS 163840" 2>&1 | grep -v amdgpu.ids > $L/s2_placement.log
cut -c1-250 $L/s2_placement.log
