#!/bin/bash
mkdir -p gpurun_out/r04
EAE_HIP_LIB=scratch/r04/libs/a_dump_early/libeae_hip.so timeout 300 python scratch/r04/dump_early_compare.py 1.0 2>&1 | grep -v "amdgpu.ids" | cut -c1-600 > gpurun_out/r04/s10_dump_early.log
cat gpurun_out/r04/s10_dump_early.log
