#!/bin/bash
mkdir -p gpurun_out/r04
L=gpurun_out/r04/s6_knockouts.log; : > $L
for v in a_identity x_noring x_noprob x_nodp x_noland_nofetch x_noring_noprob_nodp_noland_nofetch; do
  EAE_HUNT_SELF=1 EAE_HIP_LIB=scratch/r04/libs/$v/libeae_hip.so timeout 300 python scratch/r04/decode_hunt.py 1.0 3 "none,VALU only,MFMA only" 2>&1 | grep -v "amdgpu.ids\|LDS 163840" | cut -c1-250 >> $L
done
cat $L
