#!/bin/bash
# r04 GPU session 2: where do the failing decoder wavefronts run (placement probe), and do they fail when they own their CU's LDS?
mkdir -p gpurun_out/r04
L=gpurun_out/r04
EAE_HUNT_PLACEMENT=1 EAE_HIP_LIB=scratch/r04/libs/topup_p0/libeae_hip.so timeout 600 python scratch/r04/decode_hunt.py 1.0 3 "none,conv GEMM,VALU only,memory copy,MFMA only, LDS 163840" 2>&1 | grep -v amdgpu.ids > $L/s2_placement.log
# the form-override and hand-off tests after the move of the hooks out of the launch path
timeout 1200 python -m pytest tests/test_gpu_conv_split.py tests/test_gpu_latent.py tests/test_gpu_codec.py -q -x -m gpu 2>&1 | tail -5 > $L/s2_tests.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu 2>&1 | tail -5 >> $L/s2_tests.log
cat $L/s2_tests.log; cut -c1-260 $L/s2_placement.log
