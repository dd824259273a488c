#!/bin/bash
# A/B: the next K-step's activation fetch inside the MFMA region of the conv GEMM (EAE_Q_INLOOP = 0 / 1 / 2)
OUT=gpurun_out/${1:-r03_w}; mkdir -p $OUT
{
for v in 0 2 1; do
  echo "== EAE_Q_INLOOP=$v, Kodak batch 24"; SRC=conv_gemm_split EXTRA=-DEAE_Q_INLOOP=$v SCRIPT=gemm_forms.py bash scratch/variant.sh 24 512 768 default whole 2>&1 | grep -v amdgpu.ids
  echo "== EAE_Q_INLOOP=$v, 64 x 256x256"; SRC=conv_gemm_split EXTRA=-DEAE_Q_INLOOP=$v SCRIPT=gemm_forms.py bash scratch/variant.sh 64 256 256 default whole 2>&1 | grep -v amdgpu.ids
  if [ $v != 0 ]; then EAE_HIP_LIB=/tmp/eae_variant/libeae_hip.so python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv_split.py -m gpu -x -q 2>&1 | tail -2; fi
done
} | tee $OUT/inloop_ab.txt
