#!/bin/bash
# builds variants of coder_simd.hip into private libraries and runs the codec-modes script against each
for v in "-DEAE_VARIANT_NOPRIO -DEAE_VARIANT_NOFLY" "-DEAE_VARIANT_NOPRIO -DEAE_VARIANT_NOFLY -DEAE_VARIANT_NOLAND -DEAE_VARIANT_NOFETCH"; do
  echo "=== variant: [$v]"
  SRC=coder_simd EXTRA="$v" SCRIPT=r03_codec_modes.py bash scratch/variant.sh 24 2>&1 | grep -v amdgpu.ids
done
