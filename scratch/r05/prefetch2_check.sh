#!/bin/bash
# r05: the wave kernel's activations two K-steps ahead: parity of every form, then the shapes that use it
mkdir -p gpurun_out/r05r
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -3
timeout 200 python bench.py --height 256 --width 256 --batch 64 --no-cpu-baseline --no-dropin-surface --no-single-image 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('64x256', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: v['avg_ms'] for k, v in d['roofline']['per_kernel'].items()})"
python bench.py --no-cpu-baseline --no-dropin-surface > gpurun_out/r05r/bench.json 2> gpurun_out/r05r/bench.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r05r/bench.json") if l.startswith("{")][-1])
print("value", d["value"], "one-stream", d["one_stream_leg"]["value"], "roofline", d["roofline"]["frac"], "single", d["single_image"], [(o["value"], o["roofline"]["frac"]) for o in d["other_shapes"]])
PY
