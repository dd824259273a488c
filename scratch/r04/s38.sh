#!/bin/bash
mkdir -p gpurun_out/r04
{
for i in 1 2; do
echo "== latent stage, shipped"; timeout -k 10 120 python scratch/latent_timing.py < /dev/null 2>&1 | grep -v amdgpu.ids
echo "== latent stage, without the mid forms (-DEAE_NO_MID_FORMS on latent.hip)"; EAE_HIP_LIB=$PWD/scratch/r04/libs/latent_nomid/libeae_hip.so timeout -k 10 120 python scratch/latent_timing.py < /dev/null 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r04/s38_latent_ab.log 2>&1
cat gpurun_out/r04/s38_latent_ab.log
bash scratch/r04/final.sh r04/s38_final 2>&1 | tail -60 > gpurun_out/r04/s38_final.log; head -12 gpurun_out/r04/s38_final.log
