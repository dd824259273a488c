#!/bin/bash
# Builds ONE kernel file with extra flags into a private copy of libeae_hip.so and runs a script against it (EAE_HIP_LIB):
#   SRC=tconv3 EXTRA="-DEAE_T3_TRACE" SCRIPT=t3_trace.py bash scratch/variant.sh [script args]
#   SRC=tconv3 EXTRA="-DEAE_T3_RING=6" SCRIPT=t3_time.py bash scratch/variant.sh 24 512 768
#   SRC=latent EXTRA="-DEAE_LATENT_TRACE" SCRIPT=latent_trace.py bash scratch/variant.sh
set -e
cd "$(dirname "$0")/.."
C=autoencoder_based_image_compression_amd/csrc
SRC=${SRC:-tconv3}
D=/tmp/eae_variant
mkdir -p $D
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Iinclude -I$C/hip"
OBJS=$(ls build/hip/*.o)
NEW=""
for f in $SRC; do                      # SRC may name several files: the same EXTRA flags go to each
  /opt/rocm/bin/hipcc $FLAGS $EXTRA -c -o $D/$f.o $C/hip/$f.hip
  OBJS=$(echo "$OBJS" | grep -v "/$f.o")
  NEW="$NEW $D/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libeae_hip.so $OBJS $NEW
EAE_HIP_LIB=$D/libeae_hip.so python scratch/${SCRIPT:-t3_trace.py} "$@"
