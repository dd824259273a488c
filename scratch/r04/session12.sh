#!/bin/bash
mkdir -p gpurun_out/r04
./scratch/r04/libs/probe_shift64 48 20000 3 > gpurun_out/r04/s12_probe_shift.log 2>&1
EAE_HIP_LIB=scratch/r04/libs/a_shift15/libeae_hip.so timeout 300 python scratch/r04/decode_hunt.py 1.0 4 "none,VALU only,MFMA only,conv GEMM" 2>&1 | grep -v "amdgpu.ids\|LDS" | cut -c1-250 > gpurun_out/r04/s12_shift15.log
cat gpurun_out/r04/s12_probe_shift.log gpurun_out/r04/s12_shift15.log
