#!/bin/bash
# round 6: which of the two staged passes moves conv2 in the launch-by-launch leg: four builds interleaved on one box
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16
for rep in 1 2; do
for v in none binonly debonly current; do
  if [ $v = current ]; then unset EAE_HIP_LIB; else export EAE_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r06/libeae_hip_$v.so; fi
  python bench.py --no-single-image --no-cpu-baseline --no-dropin-surface 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: v['avg_ms'] for (k, v) in d['roofline']['per_kernel'].items()})"
done; done
