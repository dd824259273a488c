#!/bin/bash
# Builds a variant of tconv3 (EXTRA="-DEAE_T3_TRACE": phase cycle counters; -DEAE_T3_NOFETCH; -DEAE_T3_RING=n ...) into a
# private library and runs SCRIPT (default t3_trace.py, which needs the tracing build) against it.
set -e
cd "$(dirname "$0")/.."
C=autoencoder_based_image_compression_amd/csrc
mkdir -p /tmp/t3trace
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Iinclude -I$C/hip"
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c -o /tmp/t3trace/tconv3.o $C/hip/tconv3.hip
OBJS=$(ls build/hip/*.o | grep -v tconv3.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/t3trace/libeae_hip.so $OBJS /tmp/t3trace/tconv3.o
EAE_HIP_LIB=/tmp/t3trace/libeae_hip.so python scratch/${SCRIPT:-t3_trace.py} "$@"
