#!/bin/bash
# round 6: the two serial cores without the per-round wait for their own stores: parity (coder tests, codec tests), one image at a time, the headline
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_coder_device.py tests/test_gpu_codec.py tests/test_gpu_configs.py -x -q 2>&1 | tail -6 > gpurun_out/r06_e_tests.log
cat gpurun_out/r06_e_tests.log
timeout 300 python scratch/r06/latency.py 2>/dev/null | grep "per image" > gpurun_out/r06_e_latency.log; cat gpurun_out/r06_e_latency.log
python bench.py --no-single-image --no-cpu-baseline --no-dropin-surface 2>/dev/null | tail -1 > gpurun_out/r06_e_bench.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_e_bench.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k: v['avg_ms'] for (k, v) in d['roofline']['per_kernel'].items()})
PY
for bw in 0.125 0.05 0.0125; do python bench.py --no-single-image --no-cpu-baseline --no-dropin-surface --bin-width $bw 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bin width', $bw, d['rate_bpp'], d['value'], d['ms_per_step'], {k: v['avg_ms'] for (k, v) in d['roofline']['per_kernel'].items() if 'coder' in k})"; done
