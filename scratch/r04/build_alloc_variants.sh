#!/bin/bash
# Builds coder_simd.hip with one kernel family's reserved register moved to v63 (allocation 64) at a time -> scratch/r04/libs/alloc_<name>/
set -e
cd "$(dirname "$0")/../.."
C=autoencoder_based_image_compression_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Iinclude -I$C/hip -I$C"
OBJS=$(ls build/hip/*.o | grep -v /coder_simd.o)
build() {   # name, flags
  D=scratch/r04/libs/alloc_$1; mkdir -p $D
  /opt/rocm/bin/hipcc $FLAGS $2 -c -o $D/coder_simd.o $C/hip/coder_simd.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libeae_hip.so $OBJS $D/coder_simd.o
  rm $D/coder_simd.o
  python $C/isa_guard.py $D/libeae_hip.so | tail -1
}
build binarise "-DEAE_RES_BINARISE=63" &
build emit "-DEAE_RES_EMIT=63" &
build debinarise "-DEAE_RES_DEBINARISE=63" &
build cores "-DEAE_RES_ENCODE_CORE=63 -DEAE_RES_DECODE_CORE=63" &
build all64 "-DEAE_RES_BINARISE=63 -DEAE_RES_EMIT=63 -DEAE_RES_DEBINARISE=63 -DEAE_RES_ENCODE_CORE=63 -DEAE_RES_DECODE_CORE=63" &
build emit32 "-DEAE_RES_EMIT=31" &
build emit40 "-DEAE_RES_EMIT=39" &
wait
