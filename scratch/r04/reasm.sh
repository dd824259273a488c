#!/bin/bash
# r04: build coder_simd.hip with EXTRA flags, let EDIT (a command that rewrites the device assembly file given as $1) change the
# gfx950 assembly by hand, reassemble, and link a private libeae_hip.so into scratch/r04/libs/NAME/ (the ISA-level bisection of the
# decoder-core fault):   bash scratch/r04/reasm.sh NAME "EXTRA" 'EDIT command'
set -e
cd "$(dirname "$0")/../.."
NAME=$1; EXTRA=$2; EDIT=$3
C=autoencoder_based_image_compression_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Iinclude -I$C/hip"
W=/tmp/reasm_$NAME; rm -rf $W; mkdir -p $W
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c --save-temps=obj -o $W/coder_simd.o $C/hip/coder_simd.hip
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c --save-temps=obj -### -o $W/coder_simd.o $C/hip/coder_simd.hip 2>&1 | grep '^ "' > $W/cmds.txt
S=$W/coder_simd-hip-amdgcn-amd-amdhsa-gfx950.s
cp $S $W/original.s
eval "$EDIT $S"
diff $W/original.s $S > $W/edit.diff || true
echo "[$NAME] assembly lines changed: $(grep -c '^[<>]' $W/edit.diff || true)"
# device: assemble, link, bundle (commands 4-6 of the driver's list); host: re-embed the bundle (commands 8-10)
for i in 4 5 6 8 9 10; do eval "$(sed -n ${i}p $W/cmds.txt)"; done
D=scratch/r04/libs/$NAME; mkdir -p $D
OBJS=$(ls build/hip/*.o | grep -v "/coder_simd.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libeae_hip.so $OBJS $W/coder_simd.o
cp $W/edit.diff $D/edit.diff
echo "built $D/libeae_hip.so"
