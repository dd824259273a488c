#!/bin/bash
# round 6: the conv GEMM's epilogue with a deeper gamma ring (5, 6 against the shipped 4; registers 161 / 164 of the 168 three waves allow): parity of the
# conv kernels, then the launch-by-launch leg beside the coder and alone, and the headline, builds interleaved on one box
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16
for v in epi5 epi6; do EAE_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r06/libeae_hip_$v.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv_split.py -x -q 2>&1 | tail -1; done
for rep in 1 2; do
for v in current epi5 epi6; do
  if [ $v = current ]; then unset EAE_HIP_LIB; else export EAE_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r06/libeae_hip_$v.so; fi
  python bench.py --no-single-image --no-cpu-baseline --no-dropin-surface 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; keep=('conv2_gdn2','tconv1_igdn5','tconv2_igdn6')
print('$v', d['value'], r['frac'], r['transforms_alone']['frac'], 'beside', {k: r['per_kernel'][k]['avg_ms'] for k in keep}, 'alone', {k: r['transforms_alone']['per_kernel'][k]['avg_ms'] for k in keep})"
done; done
