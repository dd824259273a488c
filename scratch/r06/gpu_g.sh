#!/bin/bash
# round 6: one image at a time (untraced, 3 x 200 images each) with three builds of the two serial cores, interleaved on one box
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16 LATENCY_SPLIT=0
for rep in 1 2; do
for v in head olddec current; do
  if [ $v = current ]; then unset EAE_HIP_LIB; else export EAE_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r06/libeae_hip_$v.so; fi
  echo "== $v"; timeout 300 python scratch/r06/latency.py 2>/dev/null | grep "per image" | cut -c1-110
done; done
