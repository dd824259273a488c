#!/bin/bash
mkdir -p gpurun_out/r04
EAE_HUNT_PLACEMENT=1 EAE_HIP_LIB=scratch/r04/libs/a_place/libeae_hip.so timeout 300 python scratch/r04/decode_hunt.py 1.0 2 "none,VALU only,MFMA only" 2>&1 | grep -v "amdgpu.ids" | cut -c1-250 > gpurun_out/r04/s7_place.log
grep -v "LDS 163840" gpurun_out/r04/s7_place.log
