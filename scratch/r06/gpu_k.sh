#!/bin/bash
# round 6: the tree with the new encoder core, the staged debinarise pass for small steps and the early publication: the whole GPU suite on both
# libraries, one image at a time, the headline twice
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16 LATENCY_SPLIT=0
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
EAE_HIP_LIB=test python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python scratch/r06/latency.py 2>/dev/null | grep "per image" | cut -c1-110
for i in 1 2; do python bench.py --no-single-image --no-cpu-baseline --no-dropin-surface 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: v['avg_ms'] for (k, v) in d['roofline']['per_kernel'].items()})"; done
