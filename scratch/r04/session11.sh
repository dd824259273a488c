#!/bin/bash
mkdir -p gpurun_out/r04
: > gpurun_out/r04/s11_steps.log
for m in LBB3_59 LBB3_62 LBB3_65 LBB3_68 LBB3_71 LBB3_74 LBB3_77 LBB3_80; do
  echo "##### dump on entering $m" >> gpurun_out/r04/s11_steps.log
  EAE_HIP_LIB=scratch/r04/libs/d_$m/libeae_hip.so timeout 300 python scratch/r04/dump_early_compare.py 1.0 2>&1 | grep -v "amdgpu.ids" | cut -c1-700 >> gpurun_out/r04/s11_steps.log
done
grep "#####\|waves in which\|not deterministic\|===" gpurun_out/r04/s11_steps.log
