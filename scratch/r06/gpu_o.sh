#!/bin/bash
# round 6: the small-layer conv GEMM form (one wave per SIMD) with a deeper weight ring: the kernels of one image, one image at a time
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16 LATENCY_SPLIT=0
for v in "$@"; do
  if [ $v = current ]; then unset EAE_HIP_LIB; else export EAE_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r06/libeae_hip_$v.so; fi
  echo "== $v"
  bash scratch/r04/single_latency.sh r06o_$v 2>&1 | grep "conv_gemm_wave\|conv1_kernel\|gdn_kernel\|one image"
  timeout 300 python scratch/r06/latency.py 2>/dev/null | grep "per image" | cut -c1-75
done
