#!/bin/bash
# round 6: does a conv GEMM wave of at most 152 registers (three of them AND a 56-register chain wave fit a SIMD) lose less beside the coder?
# conv_3 (no epilogue) gets there with a weight ring of 6 instead of 8; the launch-by-launch leg with and without the coder, builds interleaved
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16
for rep in 1 2; do
for v in current ring6; do
  if [ $v = current ]; then unset EAE_HIP_LIB; else export EAE_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r06/libeae_hip_$v.so; fi
  python bench.py --no-single-image --no-cpu-baseline --no-dropin-surface 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', d['value'], 'beside', {k: v['avg_ms'] for (k, v) in r['per_kernel'].items() if not k.startswith('coder')}, 'alone', {k: v['avg_ms'] for (k, v) in r['transforms_alone']['per_kernel'].items()})"
done; done
