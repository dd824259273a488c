#!/bin/bash
# round 6: coder core variants: parity of the coder, then the kernels of one image (trace) and the coder's launches in the batch-24 step
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_coder_device.py -x -q 2>&1 | tail -3
bash scratch/r04/single_latency.sh ${1:-r06f} 2>&1 | grep "one image\|encode_core\|decode_core\|binarise\|sum of"
python bench.py --no-single-image --no-cpu-baseline --no-dropin-surface 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'], {k: v['avg_ms'] for (k, v) in d['roofline']['per_kernel'].items() if 'coder' in k})"
python bench.py --no-single-image --no-cpu-baseline --no-dropin-surface --bin-width 0.05 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2 bpp', d['value'], d['ms_per_step'], {k: v['avg_ms'] for (k, v) in d['roofline']['per_kernel'].items() if 'coder' in k})"
