#!/bin/bash
# r04: private builds of libeae_hip.so with variants of ONE kernel file (never the shipped library), into scratch/r04/libs/<name>/:
#   bash scratch/r04/build_variants.sh name "SRC" "-Dflags" [name "SRC" "-Dflags" ...]
set -e
cd "$(dirname "$0")/../.."
C=autoencoder_based_image_compression_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Iinclude -I$C/hip"
while [ $# -ge 3 ]; do
  NAME=$1; SRC=$2; EXTRA=$3; shift 3
  D=scratch/r04/libs/$NAME; mkdir -p $D
  OBJS=$(ls build/hip/*.o); NEW=""
  for f in $SRC; do
    /opt/rocm/bin/hipcc $FLAGS $EXTRA -c -o $D/$f.o $C/hip/$f.hip
    OBJS=$(echo "$OBJS" | grep -v "/$f.o"); NEW="$NEW $D/$f.o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libeae_hip.so $OBJS $NEW
  rm -f $D/*.o
  echo "built $D/libeae_hip.so [$EXTRA]"
done
