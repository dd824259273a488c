#!/bin/bash
# round 6: the new encoder core + early publication of the analysis side's blocks: parity of codec and coder, one image at a time
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16 LATENCY_SPLIT=0
timeout 900 python -m pytest tests/test_coder_device.py tests/test_gpu_codec.py tests/test_gpu_configs.py tests/test_gpu_surface.py -x -q 2>&1 | tail -4
for rep in 1 2; do
for v in "EAE_EARLY_PUBLISH=0" "EAE_EARLY_PUBLISH=1"; do
  echo "== $v"; env $v timeout 300 python scratch/r06/latency.py 2>/dev/null | grep "per image" | cut -c1-110
done; done
