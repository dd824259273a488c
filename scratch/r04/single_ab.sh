#!/bin/bash
# single-image latency, kernel by kernel, shipped library against variants:  bash scratch/r04/single_ab.sh tag variant [variant ...]
TAG=$1; shift
for v in shipped "$@"; do
  if [ $v = shipped ]; then unset EAE_HIP_LIB; else export EAE_HIP_LIB=$PWD/scratch/r04/libs/$v/libeae_hip.so; fi
  echo "== $v"; bash scratch/r04/single_latency.sh ${TAG}_$v 2>&1 | grep -v "^\[" | grep "one image\|conv_gemm\|gdn_kernel\|conv1\|tconv3\|latent"
done
