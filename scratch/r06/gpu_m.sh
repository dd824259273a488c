#!/bin/bash
# round 6: wave priority of the coder's kernels (s_setprio) with the new encoder core: headline and 2 bpp, builds interleaved on one box
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16
for rep in 1 2; do
for v in current prio0 prio1 prio2; do
  if [ $v = current ]; then unset EAE_HIP_LIB; else export EAE_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r06/libeae_hip_$v.so; fi
  for bw in 1.0 0.05; do python bench.py --no-single-image --no-cpu-baseline --no-dropin-surface --bin-width $bw 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', $bw, d['value'], d['ms_per_step'], d['roofline']['frac'], {k: v['avg_ms'] for (k, v) in d['roofline']['per_kernel'].items() if k in ('conv1_gdn1','conv3','tconv3','coder_encode','coder_decode')})"; done
done; done
