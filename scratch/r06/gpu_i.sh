#!/bin/bash
# round 6: binarise / debinarise with their inputs staged through LDS: parity, the kernels of one image, one image at a time, the headline
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16 LATENCY_SPLIT=0
timeout 900 python -m pytest tests/test_coder_device.py tests/test_gpu_codec.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
bash scratch/r04/single_latency.sh ${1:-r06i} 2>&1 | grep "one image\|encode_core\|decode_core\|binarise\|sum of"
timeout 300 python scratch/r06/latency.py 2>/dev/null | grep "per image" | cut -c1-110
for i in 1 2; do python bench.py --no-single-image --no-cpu-baseline --no-dropin-surface 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: v['avg_ms'] for (k, v) in d['roofline']['per_kernel'].items()})"; done
python bench.py --no-single-image --no-cpu-baseline --no-dropin-surface --bin-width 0.05 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2 bpp', d['value'], d['ms_per_step'], {k: v['avg_ms'] for (k, v) in d['roofline']['per_kernel'].items() if 'coder' in k})"
