#!/bin/bash
mkdir -p gpurun_out/r04
EAE_HIP_LIB=scratch/r04/libs/a_dump/libeae_hip.so timeout 300 python scratch/r04/dump_compare.py 1.0 2>&1 | grep -v "amdgpu.ids" | cut -c1-420 > gpurun_out/r04/s8_dump.log
cat gpurun_out/r04/s8_dump.log
