#!/bin/bash
# A/B of the SIMD coder's wave priority (EAE_SIMD_PRIO 3 vs 0): two builds of libeae_hip.so, same bench lines
for lib in prio3 prio0; do
  if [ $lib = prio0 ]; then cp autoencoder_based_image_compression_amd/lib/libeae_hip.so /tmp/keep.so; cp scratch/libeae_hip_prio0.so autoencoder_based_image_compression_amd/lib/libeae_hip.so; fi
  for bw in 1.0 0.25; do
    echo -n "$lib "; timeout 120 python bench.py --steps 20 --warmup 5 --bin-width $bw --no-cpu-baseline --no-single-image 2>/dev/null | python scratch/coder_sweep_line.py 3
  done
  if [ $lib = prio0 ]; then cp /tmp/keep.so autoencoder_based_image_compression_amd/lib/libeae_hip.so; fi
done
