#!/bin/bash
# r05: one Kodak image per step under the small-layer forms
for v in "X=1" "EAE_HIP_PACK=1" "EAE_HIP_PACK=1 EAE_HIP_FORCE_TILE=128" "EAE_HIP_FORCE_TILE=64" "EAE_HIP_PACK=1 EAE_HIP_FORCE_NT=2" "EAE_HIP_PACK=0 EAE_HIP_FORCE_NT=2" "X=1"; do
  env $v timeout 120 python scratch/r05/single_forms.py 2>&1 | grep -v amdgpu.ids | tail -1
done
echo "--- conv_3 of 64 x 256x256, per-wave placement, one-wave blocks (EAE_HIP_PACK=0) then the packed form (default)"
EAE_HIP_PACK=0 timeout 200 python scratch/r05/conv3_stamps.py 2>&1 | grep -v amdgpu.ids
timeout 200 python scratch/r05/conv3_stamps.py 2>&1 | grep -v amdgpu.ids
