#!/bin/bash
# A/B of the conv GEMM step bookkeeping (tap table through v_readlane against a scalar load per K-step) and of the weight ring
# depth of the channel-split small-layer form, in bursts (scratch/gemm_forms.py), plus the parity tests of every form.
OUT=gpurun_out/${1:-r03_l}; mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv_split.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_conv.txt
{
echo "== new (readlane), Kodak batch 24"; python scratch/gemm_forms.py 24 512 768 default whole wave 2>&1 | grep -v amdgpu.ids
echo "== old (scalar load), Kodak batch 24"; SRC=conv_gemm_split EXTRA=-DEAE_Q_TAP_SMEM SCRIPT=gemm_forms.py bash scratch/variant.sh 24 512 768 default whole wave 2>&1 | grep -v amdgpu.ids
echo "== new (readlane), 64 x 256x256"; python scratch/gemm_forms.py 64 256 256 default whole wave32_nt2 2>&1 | grep -v amdgpu.ids
echo "== old (scalar load), 64 x 256x256"; SRC=conv_gemm_split EXTRA=-DEAE_Q_TAP_SMEM SCRIPT=gemm_forms.py bash scratch/variant.sh 64 256 256 default whole wave32_nt2 2>&1 | grep -v amdgpu.ids
echo "== wave kernel, weight ring 16 deep for NT < 4, 64 x 256x256"; SRC=conv_gemm EXTRA=-DEAE_WAVE_RING_SMALL=16 SCRIPT=gemm_forms.py bash scratch/variant.sh 64 256 256 default wave32_nt2 wave32_nt1 2>&1 | grep -v amdgpu.ids
echo "== the same, Kodak batch 1"; SRC=conv_gemm EXTRA=-DEAE_WAVE_RING_SMALL=16 SCRIPT=gemm_forms.py bash scratch/variant.sh 1 512 768 default wave32_nt2 2>&1 | grep -v amdgpu.ids
echo "== ring 8, Kodak batch 1"; python scratch/gemm_forms.py 1 512 768 default wave32_nt2 2>&1 | grep -v amdgpu.ids
} | tee $OUT/gemm_ab.txt
