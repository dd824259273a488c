#!/bin/bash
# round 6: decoder core scheduling variants: parity of the coder, the decode core of one image (trace), one image at a time
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16 LATENCY_SPLIT=0
for v in "$@"; do
  if [ $v = current ]; then unset EAE_HIP_LIB; else export EAE_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r06/libeae_hip_$v.so; fi
  echo "== $v"
  timeout 600 python -m pytest tests/test_coder_device.py -x -q 2>&1 | tail -1
  bash scratch/r04/single_latency.sh r06l_$v 2>&1 | grep "encode_core\|decode_core"
  timeout 300 python scratch/r06/latency.py 2>/dev/null | grep "per image" | cut -c1-75
done
