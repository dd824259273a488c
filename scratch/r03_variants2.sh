#!/bin/bash
for v in "-DEAE_VARIANT_NOFLY" "-DEAE_VARIANT_NOPRIO -DEAE_VARIANT_NOFLY"; do
  for bw in 1.0 0.125 0.0125; do
    echo "=== variant: [$v] bw $bw"
    SRC=coder_simd EXTRA="$v" SCRIPT=r03_decode_under_load.py bash scratch/variant.sh 10 $bw 2>&1 | grep "GEMM load\|rror" | cut -c1-220
  done
done
